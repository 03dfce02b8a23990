import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device and the built product library: on a box without them a
    plain `pytest tests` skips them (the product itself never falls back -- it raises)."""
    if not any("gpu" in it.keywords for it in items):
        return
    why = None
    try:
        from falcon_amd.lib import load
        if load().fa_device_count() <= 0:
            why = "no HIP device visible"
    except Exception as e:  # the .so is missing or does not load
        why = "libfalcon_amd.so unavailable: %s" % (e,)
    if why:
        skip = pytest.mark.skip(reason=why)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


def load_golden(name):
    with gzip.open(os.path.join(GOLDEN, name + ".json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def port():
    from oracle.pyoracle import Port, build
    build()
    return Port()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/falcon_ref.so not built (needs /root/reference)")
    return Ref()
