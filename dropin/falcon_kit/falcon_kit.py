"""``falcon_kit.falcon_kit`` on libfalcon_amd.so (reference: falcon_kit/falcon_kit.py)."""
from falcon_amd.falcon_kit import *  # noqa: F401,F403
from falcon_amd.falcon_kit import consensus_of  # noqa: F401
