"""The CPU oracle (oracle/falcon_oracle.c) against the golden vectors produced by
the compiled reference (oracle/gen_golden.py).  This is what pins the oracle."""
import pytest

from conftest import load_golden
from helpers import check_align_case, check_config_case, check_hits_case, check_pile_case

F1 = load_golden("f1_f2_hits_ranges")["cases"]
F2X = load_golden("f2_ranges_extra")["cases"]
F3 = load_golden("f3_align")["cases"]
F4 = load_golden("f4_piles")["cases"]
F8 = load_golden("f8_configs")["cases"]


@pytest.mark.parametrize("case", F1, ids=[c["name"] for c in F1])
def test_hits_and_ranges(port, case):
    check_hits_case(port, case)


@pytest.mark.parametrize("case", F2X, ids=[c["name"] for c in F2X])
def test_ranges_extra(port, case):
    assert list(port.best_range(case["q"], case["t"], case["bin"], case["th"])) == case["range"]


@pytest.mark.parametrize("case", F3, ids=[c["name"] for c in F3])
def test_align(port, case):
    check_align_case(port, case)


@pytest.mark.parametrize("case", F4, ids=[c["name"] for c in F4])
def test_piles(port, case):
    check_pile_case(port, case)


@pytest.mark.parametrize("case", F8, ids=[c["name"] for c in F8])
def test_full_size_config_piles(port, case):
    """SURVEY.md 8d configs 2, 4 (the 200-read cap binds) and 5 (two haplotypes) at full
    size: the reference's consensus of the canonical pile."""
    check_config_case(port, case)


def test_golden_quirks_are_present():
    """The fixtures really exercise Q1-Q3 (first base dropped, trailing 'A', tail loss)."""
    c = {x["name"]: x for x in F4}["identical_copies_12"]
    seed = c["seqs"][0]
    assert c["sequence"][:-1] == seed[1:len(c["sequence"])]
    assert len(c["sequence"]) == 1987 and c["sequence"][-1] == "A"
    low = {x["name"]: x for x in F4}["identical_copies_3_lowcov"]
    assert low["sequence"].islower()
