"""The real command line on the GPU: stdin -> stdout byte parity with the golden
vectors produced by the reference's own driver (tests/golden/f5_cli, f6_cli_trim),
through the module path, the drop-in ``falcon_kit`` overlay and the console script."""
import os
import subprocess
import sys

import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F5 = load_golden("f5_cli")
F6 = load_golden("f6_cli_trim")


def run_cmd(cmd, stdin_text, extra_path=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(([extra_path] if extra_path else []) + [ROOT])
    p = subprocess.run(cmd, input=stdin_text, capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("case", F5["runs"] + F6["runs"],
                         ids=[" ".join(r["argv"]) or "defaults" for r in F5["runs"] + F6["runs"]])
def test_module_cli(case):
    out = run_cmd([sys.executable, "-m", "falcon_amd.mains.consensus"] + case["argv"] +
                  ["--n-core", "1"], F5["stdin"])
    assert out == case["stdout"]


F11 = load_golden("f11_cli_config1")


def _config1_stream():
    """The LA4Falcon text of BASELINE config 1, rebuilt the way oracle/gen_golden.py f11 built
    it: the t1.fa read (frozen as the seed of f4's `t1_config1` pile) and 20 noisy copies of
    it from the same seeded generator; the fixture's digest pins the bytes."""
    import hashlib
    import numpy as np
    from falcon_amd.synth import noisy, pile_to_la4falcon
    seed = {c["name"]: c for c in load_golden("f4_piles")["cases"]}["t1_config1"]["seqs"][0]
    codes = np.frombuffer(seed.encode(), dtype=np.uint8)
    t1c = np.zeros(len(codes), dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        t1c[codes == ch] = i
    r1 = np.random.default_rng(1)
    derived = [noisy(t1c, r1, 0.12) for _ in range(20)]
    text = pile_to_la4falcon(F11["seed_id"], t1c, derived, 1) + "- -\n"
    assert hashlib.sha1(text.encode()).hexdigest() == F11["stdin_sha"]
    return text


@pytest.mark.parametrize("case", F11["runs"], ids=[" ".join(r["argv"]) or "defaults" for r in F11["runs"]])
def test_config1_t1_pile_cli(case):
    """BASELINE config 1 at the CLI level: the test_data/t1.fa-derived pile as LA4Falcon text
    through the worker, byte for byte what the reference's own driver prints for it
    (tests/golden/f11_cli_config1, oracle/gen_golden.py f11)."""
    out = run_cmd([sys.executable, "-m", "falcon_amd.mains.consensus"] + case["argv"] + ["--n-core", "1"],
                  _config1_stream())
    assert out == case["stdout"]


def test_dropin_overlay_and_console_script():
    case = F5["runs"][1]  # the fc_run_ecoli.cfg flags
    out = run_cmd([sys.executable, "-m", "falcon_kit.mains.consensus"] + case["argv"] +
                  ["--n-core", "1"], F5["stdin"], extra_path=os.path.join(ROOT, "dropin"))
    assert out == case["stdout"]
    out = run_cmd([os.path.join(ROOT, "bin", "fc_consensus")] + case["argv"] + ["--n-core", "1"],
                  F5["stdin"])
    assert out == case["stdout"]


def test_overlay_exposes_reference_names():
    code = ("import falcon_kit, sys; from falcon_kit import kup, DWA, falcon, ConsensusData;"
            "import falcon_kit.falcon_kit as fk;"
            "print(fk.consensus_of(['ACGT'*300]*12, 4, 8, 0.7)[:20])")
    out = run_cmd([sys.executable, "-c", code], "", extra_path=os.path.join(ROOT, "dropin"))
    assert out.strip() == ("ACGT" * 300)[1:21]


def test_a_deep_pile_is_corrected_like_any_other():
    """--max-n-read 2000 lets a pile of ~1200 usable reads through (rounds 1-3 reported piles of
    more than 1023 and left them uncorrected, exit status 3): it is printed where it belongs, with
    the bytes the reference's own C makes of it (falcon.c:597-647 loops over any n_seq)."""
    from falcon_amd.mains.consensus import fasta_records
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_la4falcon, pile_to_seqs
    from oracle.pyoracle import Port, Ref, have_ref
    chunks = []
    for i in range(3):
        seed, rd = make_pile(1300 + i, S=2500, coverage=14, min_read=500, mean_read=1500, sd_read=400)
        chunks.append(pile_to_la4falcon("%09d" % i, seed, rd, 100000 * i + 1))
    seed, rd = make_pile(1310, S=2500, coverage=830, e=0.08, min_read=1500, mean_read=2200, sd_read=200)
    deep = pile_to_la4falcon("%09d" % 7, seed, rd, 700001)
    opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "2000", "--n-core", "1"]
    cmd = [sys.executable, "-m", "falcon_amd.mains.consensus"] + opts
    clean = run_cmd(cmd, "".join(chunks) + "- -\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    text = chunks[0] + deep + chunks[1] + chunks[2] + "- -\n"
    p = subprocess.run(cmd, input=text, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    pile = [codes_to_str(x) for x in pile_to_seqs(seed, rd, 2000)]
    assert len(pile) > 1150
    cns = (Ref() if have_ref() else Port()).generate_consensus(pile, 4, 8, 0.70)[0]
    want = "".join(fasta_records("%09d" % 7, cns, False, True))
    first = run_cmd(cmd, chunks[0] + "- -\n")
    assert clean.startswith(first) and len(want) > 2000
    assert p.returncode == 0, p.stderr[-400:]
    assert p.stdout == first + want + clean[len(first):]


def test_a_dirty_pile_in_the_stream_fails_alone():
    """A read with an N (or any byte that is not upper-case ACGT) in the middle of a stream:
    its pile is named on stderr and left out, every other pile is printed exactly as without
    it, exit status 3 -- the stream does not die (the reference aligns raw characters there;
    LA4Falcon never emits them)."""
    from falcon_amd.synth import make_pile, pile_to_la4falcon
    chunks = []
    for i in range(4):
        seed, rd = make_pile(1700 + i, S=2500, coverage=14, min_read=500, mean_read=1500, sd_read=400)
        chunks.append(pile_to_la4falcon("%09d" % i, seed, rd, 100000 * i + 1))
    lines = chunks[2].split("\n")
    name, bases = lines[3].split(" ")
    lines[3] = name + " " + bases[:100] + "N" + bases[101:]
    dirty = "\n".join(lines)
    opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--n-core", "1"]
    cmd = [sys.executable, "-m", "falcon_amd.mains.consensus"] + opts
    clean = run_cmd(cmd, chunks[0] + chunks[1] + chunks[3] + "- -\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for extra in ({}, {"FALCON_AMD_BATCH_BASES": "60000"}, {"FALCON_AMD_PY_READER": "1"}):
        p = subprocess.run(cmd, input=chunks[0] + chunks[1] + dirty + chunks[3] + "- -\n", capture_output=True,
                           text=True, env=dict(env, **extra), cwd=ROOT, timeout=600)
        assert p.returncode == 3 and p.stdout == clean, extra
        assert "seed 000000002 is not corrected" in p.stderr and "holds byte 0x4e at position 100" in p.stderr


def test_native_and_python_printers_agree():
    """The worker formats a batch's FASTA text in native code (fa_batch_fasta);
    FALCON_AMD_PY_PRINTER=1 takes the per-pile python path (fasta_records, pinned to the
    reference's driver by f5_cli): same bytes in the three output modes, also around a pile
    that fails alone, with batches of a few piles each."""
    from falcon_amd.synth import make_pile, pile_to_la4falcon
    chunks = []
    for i in range(7):
        seed, rd = make_pile(1500 + i, S=2400 + 300 * i, coverage=16, e=0.10 + 0.01 * i, min_read=500,
                             mean_read=1500, sd_read=400)
        chunks.append(pile_to_la4falcon("%09d" % i, seed, rd, 100000 * i + 1))
    seed, rd = make_pile(1310, S=2500, coverage=20, e=0.08, min_read=1500, mean_read=2200, sd_read=200)
    lines = pile_to_la4falcon("%09d" % 77, seed, rd, 700001).split("\n")
    name, bases = lines[3].split(" ")
    lines[3] = name + " " + bases[:100] + "N" + bases[101:]  # (the pile that fails alone: a byte outside ACGT)
    chunks.insert(3, "\n".join(lines))
    text = "".join(chunks) + "- -\n"
    env = dict(os.environ, PYTHONPATH=ROOT, FALCON_AMD_BATCH_BASES="150000")
    for mode in ([], ["--output-multi"], ["--output-full"]):
        cmd = [sys.executable, "-m", "falcon_amd.mains.consensus"] + mode + \
              ["--min-idt", "0.70", "--min-cov", "1", "--max-n-read", "2000", "--n-core", "1"]
        nat = subprocess.run(cmd, input=text, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        py = subprocess.run(cmd, input=text, capture_output=True, text=True, cwd=ROOT, timeout=600,
                            env=dict(env, FALCON_AMD_PY_PRINTER="1"))
        assert nat.returncode == 3 and py.returncode == 3
        assert nat.stdout == py.stdout and nat.stdout.count(">") >= 7
        assert "seed 000000077 is not corrected" in nat.stderr and "seed 000000077 is not corrected" in py.stderr


def test_length_limits_through_the_command_line():
    """Sequences of more than 100 000 bases are cut to 99 999 (consensus.py:162,178-179) and
    the consensus core takes seeds below 100 000 (falcon.c:343): a pile on a 99.3 kb seed
    with a 101 kb read (cut by the reader), and a pile whose seed line itself is 100.6 kb
    (cut to 99 999, the largest seed there can be) -- stdout equals the restated parser +
    the CPU oracle + the output rules on the same text."""
    import io
    import numpy as np
    from falcon_amd.mains import consensus as cli
    from falcon_amd.synth import codes_to_str, noisy
    from oracle.pyoracle import Port
    rng = np.random.default_rng(99)
    lines, seeds = [], []
    for pile, (window, seed_e) in enumerate(((95000, 0.13), (98500, 0.05))):
        genome = rng.integers(0, 4, 104000).astype(np.uint8)
        seeds.append(codes_to_str(noisy(genome[:window], rng, seed_e)))
        lines.append("%08d %s" % (pile, seeds[-1]))
        k = 1
        for start in range(0, window - 20000, 9000):
            lines.append("%08d %s" % (1000 + 100 * pile + k,
                                      codes_to_str(noisy(genome[start:start + 31000], rng, 0.12))))
            k += 1
        lines.append("%08d %s" % (1000 + 100 * pile + k, codes_to_str(noisy(genome[:100500], rng, 0.05))))
        lines.append("+ +")
    text = "\n".join(lines) + "\n- -\n"
    assert 99000 < len(seeds[0]) < 100000 and len(seeds[1]) > 100000
    assert max(len(ln) for ln in lines) > 100000 + 9
    opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "2", "--min-cov-aln", "2", "--min-n-read", "3",
            "--n-core", "1"]
    out = run_cmd([sys.executable, "-m", "falcon_amd.mains.consensus"] + opts, text)
    args = cli.parse_args(["prog"] + opts)
    cfg = cli.settings_from(args)
    port, want = Port(), []
    piles = list(cli.PileReader(io.StringIO(text), cfg, args.min_n_read, args.min_len_aln))
    assert [len(p[0]) for _, p in piles] == [len(seeds[0]), 99999]
    assert max(len(r) for _, p in piles for r in p) == 99999
    for sid, pile in piles:
        want.append(cli.fasta_records(sid, port.generate_consensus(pile, 2, 8, 0.70)[0], False, True))
    assert out == "".join(want) and out.count(">") >= 2 and len(out) > 150000


@pytest.mark.gpu
def test_two_jobs_started_together_share_what_there_is():
    """bench.py's N > 1 end-to-end leg (benchlib/e2e.py end_to_end_workers): single-stream jobs
    started at the same moment, each taking a GPU through the lock slots -- on a one-GPU box both
    land on device 0, and both print the same FASTA."""
    sys.path.insert(0, ROOT)
    from benchlib.e2e import end_to_end_workers
    from benchlib.workloads import gen_piles
    piles = gen_piles(range(40, 46), 1, dict(S=3000, coverage=20.0, het=0.0))
    out = end_to_end_workers(piles, 2, repeats=3)
    assert out["every_fasta_identical"] and len(out["workers"]) == 2, out
    import torch
    assert out["distinct_devices"] == min(2, torch.cuda.device_count()), out
    for w in out["workers"]:
        assert w["devices"] is not None and w["wall_s"] > 0, out


@pytest.mark.gpu
def test_the_real_engine_under_world_size_2_on_one_gpu():
    """`bench.py --gpus N` is one process per GPU, the ranks lined up and the measurement
    reduced through torch.distributed.  A box with one GPU cannot run RCCL with two ranks (it refuses two
    ranks on one device), but it can run everything else of that code path on the real engine:
    FALCON_BENCH_BACKEND=gloo + FALCON_BENCH_ONE_DEVICE=1 put both ranks on device 0.  The line must
    say two ranks, carry both ranks' own clocks, and start two single-stream workers in its
    end-to-end leg."""
    import json
    env = dict(os.environ, FALCON_BENCH_BACKEND="gloo", FALCON_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--piles", "96", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert len(d["per_rank"]) == 2 and all(r["bases_per_sec"] > 0 for r in d["per_rank"])
    # whole-job value = the units of both ranks over the slower rank's time
    assert d["value"] <= sum(r["bases_per_sec"] for r in d["per_rank"]) * 1.001
    assert d["value"] >= 2 * min(r["bases_per_sec"] for r in d["per_rank"]) * 0.999
    if "end_to_end_workers" in d and d["end_to_end_workers"]:
        assert len(d["end_to_end_workers"]["workers"]) == 2


@pytest.mark.gpu
def test_the_real_engine_under_world_size_8_on_one_gpu():
    """The shape of the 8-GPU run nobody could launch here (the reference's fan-out:
    consensus.py:264-274, consensus_split.py:55-85), on the one GPU of the box: eight ranks rendezvous over
    127.0.0.1, each with the real Engine, its resident batches and its share of the pile generators (the
    container's CPU quota divided by eight), lined up by barriers, the measurement reduced over
    torch.distributed; then eight single-stream workers started together, each a process of its own that
    takes a lock slot (all eight on device 0 here: eight different slots of it).  Eight arenas of 6.8 GB fit
    the 288 GB of one MI355X."""
    import json
    env = dict(os.environ, FALCON_BENCH_BACKEND="gloo", FALCON_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--piles", "48", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert len(d["per_rank"]) == 8 and all(r["bases_per_sec"] > 0 for r in d["per_rank"])
    assert d["value"] <= sum(r["bases_per_sec"] for r in d["per_rank"]) * 1.001
    assert d["value"] >= 8 * min(r["bases_per_sec"] for r in d["per_rank"]) * 0.999
    # the generators of all ranks together stay inside the CPU quota (8 ranks forked 32 each once)
    g = d["generators"]
    assert g["ranks"] == 8 and g["processes_per_rank"] >= 1
    assert g["processes_per_rank"] * 8 <= max(8, int(g["cpu_quota_cores"] or os.cpu_count()))
    w = d["end_to_end_workers"]
    assert len(w["workers"]) == 8 and w["every_fasta_identical"], w
    # one GPU on this box ...
    assert {x["devices"] for x in w["workers"]} == {"0"}, [(x["devices"], x["lock_slots"]) for x in w["workers"]]
    assert w["distinct_lock_slots"] == 8, w                             # ... eight different slots of it
    assert d["end_to_end"]["piles_per_sec"], d["end_to_end"]            # eight streams through the multi-stream worker


@pytest.mark.gpu
def test_a_job_served_by_the_node_s_server_prints_the_same_bytes():
    """falcon_amd.mains.consensus_server on the real engine: two jobs one after the other and two at the same
    time hand their stdin / stdout to it; each FASTA equals what the stand-alone worker prints for the stream."""
    import signal
    import tempfile
    sys.path.insert(0, ROOT)
    from benchlib.workloads import gen_piles, write_la4falcon
    piles = gen_piles(range(60, 68), 1, dict(S=4000, coverage=20.0, het=0.0))
    with tempfile.TemporaryDirectory() as tmp:
        src, sock = os.path.join(tmp, "piles.txt"), os.path.join(tmp, "srv.sock")
        with open(src, "wb") as f:
            write_la4falcon(piles, f, 2)
        cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70", "--min-cov", "4",
               "--max-n-read", "200", "--n-core", "1"]
        with open(src) as fin:
            want = subprocess.run(cmd, stdin=fin, stdout=subprocess.PIPE, check=True, cwd=ROOT, timeout=600).stdout
        assert want.count(b">") >= 8
        srv = subprocess.Popen([sys.executable, "-m", "falcon_amd.mains.consensus_server", "--socket", sock], cwd=ROOT,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            assert "ready" in srv.stdout.readline()
            env = dict(os.environ, FALCON_AMD_SERVER=sock)
            for _ in range(2):
                with open(src) as fin:
                    got = subprocess.run(cmd, stdin=fin, stdout=subprocess.PIPE, check=True, cwd=ROOT, timeout=600, env=env).stdout
                assert got == want
            ps = []
            for j in range(2):
                ps.append((subprocess.Popen(cmd, stdin=open(src), stdout=open(os.path.join(tmp, "o%d" % j), "wb"), cwd=ROOT, env=env), j))
            for p, j in ps:
                assert p.wait(timeout=600) == 0
                assert open(os.path.join(tmp, "o%d" % j), "rb").read() == want
        finally:
            srv.send_signal(signal.SIGTERM)
            srv.wait(timeout=60)
        assert not os.path.exists(sock)
