"""Overlay that answers the reference's import paths with the MI355X implementation.

Put this directory in front of the reference package on PYTHONPATH (or copy these
three files over falcon_kit/__init__.py-level imports) and

    python -m falcon_kit.mains.consensus <falcon_sense_option>

-- the command fc_run's consensus task writes (falcon_kit/mains/consensus_task.py:90)
-- runs on the GPU.  ``from falcon_kit import kup, DWA, falcon`` resolves to
libfalcon_amd.so the same way (reference: falcon_kit/__init__.py)."""
from falcon_amd.falcon_kit import *  # noqa: F401,F403
