"""Overlay that answers the reference's import paths with the MI355X implementation.

Put this directory in front of the reference package on PYTHONPATH (or copy these
three files over falcon_kit/__init__.py-level imports) and

    python -m falcon_kit.mains.consensus <falcon_sense_option>

-- the command fc_run's consensus task writes (falcon_kit/mains/consensus_task.py:90)
-- runs on the GPU.  ``from falcon_kit import kup, DWA, falcon`` resolves to
libfalcon_amd.so the same way (reference: falcon_kit/__init__.py).

Only ``falcon_kit`` itself, ``falcon_kit.falcon_kit`` and ``falcon_kit.mains.consensus`` are
answered here: ``__path__`` is extended over every other ``falcon_kit`` directory on
sys.path, so ``falcon_kit.mains.consensus_task``, ``falcon_kit.io``, ``falcon_kit.bash`` ...
-- which the consensus job imports before it writes the pipe
(falcon_kit/pype_tasks.py:38, mains/consensus_task.py:8-9) -- still come from the
reference package behind this overlay."""
__path__ = __import__("pkgutil").extend_path(__path__, __name__)
from falcon_amd.falcon_kit import *  # noqa: F401,F403
