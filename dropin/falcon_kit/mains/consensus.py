"""``python -m falcon_kit.mains.consensus`` -> the GPU worker
(reference: falcon_kit/mains/consensus.py, setup.py:51 ``fc_consensus``)."""
import sys

from falcon_amd.mains.consensus import console_main, main, parse_args, run  # noqa: F401

if __name__ == "__main__":
    console_main(sys.argv)
