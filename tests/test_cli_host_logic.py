"""The fc_consensus command line: parser, pile admission/selection and FASTA
formatting against the CLI golden vectors produced by the reference's own
driver.  The consensus calls themselves are served by the CPU oracle here (this
is a test harness; the product CLI only ever uses the GPU engine)."""
import io
import os

import pytest

from conftest import load_golden
from falcon_amd.mains import consensus as cli

F5 = load_golden("f5_cli")
F6 = load_golden("f6_cli_trim")


def oracle_map(port, min_cov, min_idt):
    def run(piles):
        for p in piles:
            yield port.generate_consensus(p, min_cov, 8, min_idt)[0]
    return run


def run_cli(argv, stdin_text, cmap):
    args = cli.parse_args(["fc_consensus"] + argv + ["--n-core", "0"])
    out = io.StringIO()
    cli.run(args, stdin=io.StringIO(stdin_text), stdout=out, consensus_map=cmap)
    return out.getvalue()


@pytest.mark.parametrize("case", F5["runs"], ids=[" ".join(r["argv"]) or "defaults" for r in F5["runs"]])
def test_cli_matches_reference_driver(port, case):
    args = cli.parse_args(["x"] + case["argv"])
    got = run_cli(case["argv"], F5["stdin"], oracle_map(port, args.min_cov, args.min_idt))
    assert got == case["stdout"]


@pytest.mark.parametrize("case", F6["runs"], ids=[" ".join(r["argv"]) for r in F6["runs"]])
def test_cli_trim_matches_reference_driver(port, case):
    args = cli.parse_args(["x"] + case["argv"])
    got = run_cli(case["argv"], F5["stdin"], oracle_map(port, args.min_cov, args.min_idt))
    assert got == case["stdout"]


def test_flags_and_defaults_match_the_reference_cli():
    a = cli.parse_args(["x"])
    assert (a.n_core, a.min_cov, a.min_cov_aln, a.max_cov_aln, a.min_len_aln, a.min_n_read,
            a.max_n_read, a.min_idt, a.edge_tolerance, a.trim_size, a.verbose_level) == \
        (24, 6, 10, 0, 0, 10, 500, 0.70, 1000, 50, 2.0)
    assert not (a.trim or a.output_full or a.output_multi)
    b = cli.parse_args("x --output-multi --min-idt 0.70 --min-cov 4 --max-n-read 200 --n-core 6".split())
    assert b.output_multi and b.min_cov == 4 and b.max_n_read == 200 and b.n_core == 6


def test_help_exits_cleanly():
    with pytest.raises(SystemExit) as e:
        cli.parse_args(["prog", "--help"])   # reference test/test_consensus.py:5-9
    assert e.value.code == 0


def test_select_reads_is_stable_and_capped():
    pile = ["S" * 10, "a" * 5, "b" * 7, "c" * 7, "d" * 3]
    assert cli.select_reads(pile, 4, 0) == ["S" * 10, "b" * 7, "c" * 7, "a" * 5]
    # coverage cap: stop once bases // seed_len exceeds the cap
    assert cli.select_reads(pile, 500, 1) == ["S" * 10, "b" * 7, "c" * 7, "a" * 5, "d" * 3][:5]
    assert len(cli.select_reads(["S" * 4] + ["r" * 4] * 9, 500, 1)) == 3


def test_pile_reader_grammar():
    text = "s1 ACGT\nr1 AAAA\nr1 CCCC\nthree tokens here\nr2 GG\n+ +\nx TTTT\n* *\ns2 AC\n- -\nz AAAA\n+ +\n"
    cfg = cli.Settings(4, 8, 500, 0.7, 1000, 50, 0, 0)
    piles = list(cli.PileReader(io.StringIO(text), cfg, 1, 0))
    assert piles == [("s1", ["ACGT", "ACGT", "AAAA", "GG"])]


def test_dropin_overlay_falls_through_to_the_reference_package(tmp_path):
    """PYTHONPATH=dropin:<reference>: the overlay answers falcon_kit, falcon_kit.falcon_kit and
    falcon_kit.mains.consensus; every other submodule (the consensus job imports
    falcon_kit.mains.consensus_task, falcon_kit.io, ... first: pype_tasks.py:38,
    consensus_task.py:8-9) must still resolve in the package behind it.  The package behind
    it here is a stand-in with the same shape."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = tmp_path / "ref" / "falcon_kit"
    (ref / "mains").mkdir(parents=True)
    (ref / "__init__.py").write_text("WHO = 'reference'\n")
    (ref / "io.py").write_text("WHO = 'reference io'\n")
    (ref / "falcon_kit.py").write_text("WHO = 'reference falcon_kit'\n")
    (ref / "mains" / "__init__.py").write_text("")
    (ref / "mains" / "consensus_task.py").write_text("from .. import io\nWHO = 'reference task ' + io.WHO\n")
    (ref / "mains" / "consensus.py").write_text("WHO = 'reference consensus'\n")
    code = ("import falcon_kit, falcon_kit.io, falcon_kit.falcon_kit, falcon_kit.mains.consensus_task as t, "
            "falcon_kit.mains.consensus as c\n"
            "print(t.WHO); print(falcon_kit.io.WHO); print(hasattr(falcon_kit, 'kup'), "
            "hasattr(falcon_kit.falcon_kit, 'kup'), hasattr(c, 'parse_args'), hasattr(c, 'WHO'))\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "dropin"), str(tmp_path / "ref"), root]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:3] == ["reference task reference io", "reference io", "True True True False"]


def test_native_output_rules_equal_the_python_ones():
    """fa_fasta_records (falcon_amd/csrc/fasta.cpp, what fa_batch_fasta applies to every pile
    of a batch) against fasta_records, which tests/golden/f5_cli pins to the reference's own
    driver (consensus.py:275-299): short strings, no solid run, ties between longest runs,
    more than ten runs, runs of exactly 499/500 and of multiples of 80, lower-case and
    foreign characters as separators."""
    import ctypes as C
    import random
    from falcon_amd.lib import load
    lib = load()
    rng = random.Random(9)

    def run_of(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    cases = ["", "ACGT" * 10, "acgt" * 300, run_of(499), run_of(500), run_of(560) + "a" + run_of(560),
             run_of(700) + "n" + run_of(800) + "c" + run_of(800) + "N-" + run_of(30),
             "a".join(run_of(500 + 80 * (i % 3)) for i in range(14)),
             "g".join(run_of(rng.choice([3, 499, 500, 501, 640, 641])) for _ in range(30)),
             run_of(100) + "t" * 450, "x" + run_of(1600) + "y", run_of(480) + "a" + run_of(480)]
    for _ in range(60):
        n_piece = rng.randint(1, 25)
        cases.append("".join(run_of(rng.choice([1, 7, 80, 160, 499, 500, 503, 1200])) +
                             rng.choice(["", "a", "ac", "N", "-", "tg"]) for _ in range(n_piece)))
    buf = C.create_string_buffer(1 << 20)
    n_out = 0
    for cns in cases:
        for mode, (full, multi) in enumerate([(False, False), (False, True), (True, False)]):
            want = cli.fasta_records("000123", cns, full, multi).encode("ascii")
            n = lib.fa_fasta_records(b"000123", cns.encode("ascii"), len(cns), mode, buf, len(buf))
            assert n == len(want) and buf.raw[:n] == want, (mode, len(cns))
            n_out += bool(want)
    assert n_out > 100
    assert lib.fa_fasta_records(b"x", b"ACGT", 4, 3, buf, 10) < 0  # unknown mode


def test_console_main_leaves_without_teardown_and_keeps_the_exit_status(tmp_path):
    """`python -m falcon_amd.mains.consensus` / bin/fc_consensus end through console_main: once run() has
    returned, output is flushed and the process leaves with os._exit (0, or 3 when piles failed alone) --
    no interpreter teardown, so nothing registered with atexit runs; main() called by a host program
    returns (or raises SystemExit(3)) as it always did; FALCON_AMD_SLOW_EXIT=1 restores that for the
    console command too."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = """
import atexit, os, sys
sys.path.insert(0, %r)
import falcon_amd.mains.consensus as c
atexit.register(lambda: sys.stderr.write("ATEXIT RAN\\n"))
def fake_run(args, leave_open=False, **kw):
    sys.stdout.write(">r\\nACGT\\n")          # (left in the buffer: _leave has to flush it)
    sys.stderr.write("leave_open=%%s\\n" %% leave_open)
    if os.environ.get("FAIL_ONE"):
        c.FAILED_PILES.append("000000007")
c.run = fake_run
getattr(c, sys.argv[1])(["prog", "--n-core", "1"])
sys.stderr.write("RETURNED\\n")
""" % ROOT

    def go(entry, **env):
        e = dict(os.environ, **env)
        e.pop("FALCON_AMD_SLOW_EXIT", None) if "FALCON_AMD_SLOW_EXIT" not in env else None
        return subprocess.run([sys.executable, "-c", prog, entry], capture_output=True, text=True, env=e, timeout=120)
    p = go("console_main")
    assert (p.returncode, p.stdout) == (0, ">r\nACGT\n") and "leave_open=True" in p.stderr
    assert "ATEXIT RAN" not in p.stderr and "RETURNED" not in p.stderr
    p = go("console_main", FAIL_ONE="1")
    assert p.returncode == 3 and p.stdout == ">r\nACGT\n" and "1 pile(s) were not corrected" in p.stderr
    p = go("console_main", FALCON_AMD_SLOW_EXIT="1")
    assert p.returncode == 0 and "leave_open=False" in p.stderr and "RETURNED" in p.stderr and "ATEXIT RAN" in p.stderr
    p = go("main")
    assert p.returncode == 0 and "leave_open=False" in p.stderr and "RETURNED" in p.stderr and "ATEXIT RAN" in p.stderr
    p = go("main", FAIL_ONE="1")
    assert p.returncode == 3 and "RETURNED" not in p.stderr and "ATEXIT RAN" in p.stderr
