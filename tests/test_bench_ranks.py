"""bench.py's N > 1 path on CPU: the rank logic (bench_rank: barrier-bracketed timing, sum of
units over ranks / max time, the ranks-that-ran check) over gloo with a stand-in engine, the
refusal to run fewer ranks than --gpus asks for, and the launcher/--gpus consistency check.
On the GPU box the same function runs over RCCL with falcon_amd.engine.Engine."""
import io
import json
import os
import socket
import subprocess
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Stats:
    def __init__(self, n_piles, bases):
        self.O, self.n_piles, self.n_seqs, self.n_aligned = bases, n_piles, 10 * n_piles, 9 * n_piles
        self.L, self.C, self.D, self.A, self.T = 1000 * n_piles, 5000 * n_piles, 100 * n_piles, 900 * n_piles, 100 * n_piles
        for n in ("ms_align", "ms_consensus", "ms_chain", "ms_index", "ms_tags", "ms_links",
                  "ms_score", "ms_backtrace"):
            setattr(self, n, 1.0)
        self.ms_total = 9.0

    def b_alg(self):
        return self.L // 4 + 4 * self.C + 8 * self.D + 16 * self.A + 12 * self.T + 5 * self.O


class _Batch:
    def __init__(self, piles):
        self.n_pile = len(piles)
        self.bases = sum(len(p[0]) for p in piles)

    def run(self, *a):
        import time
        time.sleep(0.05)  # (a step long enough for the rounded ms_per_step to be exact to 1e-3)
        return self

    def stats(self):
        return _Stats(self.n_pile, self.bases)

    def free(self):
        pass


class StandInEngine:
    def __init__(self, device):
        self.device = device

    def batch(self, piles):
        return _Batch(piles)

    def close(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank(rank, world, port, gpus, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    args = bench.parse_args(["--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--piles", "4",
                             "--no-cpu-baseline", "--no-end-to-end"])
    plumb = bench.Plumbing(rank, rank, world, backend="gloo")
    piles = [[b"A" * (100 * (rank + 1))] * 3 for _ in range(args.piles)]  # rank r: 4 piles x 100(r+1) "bases"
    out = io.StringIO()
    try:
        res = bench.bench_rank(args, plumb, StandInEngine, piles, out=out)
        q.put((rank, "ok", out.getvalue(), res is not None))
    except Exception as e:
        q.put((rank, "error", repr(e), False))
    finally:
        plumb.close()


def _run_world(world, gpus):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, gpus, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    return res


def test_two_ranks_reduce_to_one_line():
    res = _run_world(2, 2)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert res[1][2] == "" and not res[1][3]          # only rank 0 prints
    line = json.loads(res[0][2])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    # whole-job aggregate: 4 piles x 100 bases on rank 0 + 4 x 200 on rank 1, per step
    assert line["value"] * line["ms_per_step"] * 1e-3 == pytest.approx(1200.0, rel=5e-3)
    assert line["piles_per_sec"] * line["ms_per_step"] * 1e-3 == pytest.approx(8.0, rel=5e-3)
    assert "cpu_baseline" not in line and "end_to_end" not in line  # rank 0, N = 1 only


def test_rank_count_must_match_gpus():
    res = _run_world(2, 4)   # --gpus 4 but only two ranks took part
    assert res[0][1] == "error" and "2 rank(s) took part" in res[0][2]


def test_bare_gpus_n_refuses_without_devices():
    """`python bench.py --gpus 2` on a box with fewer than 2 devices fails loudly (it would
    re-exec itself under torch.distributed.run otherwise)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    if r.returncode == 0:
        pytest.skip("this box has >= 2 HIP devices")
    assert "refusing to run fewer ranks" in r.stderr


def test_launcher_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "launcher started 2 rank(s)" in r.stderr


def test_traffic_record_digest_ignores_comments_not_code():
    """`roofline.traffic` is only shown for the code it was measured on: the digest is taken of
    the kernel's sources without comments and layout -- a reworded comment keeps it, a changed
    token, or a changed string (the inline asm), does not."""
    import bench
    a = 'int f(int x) { // add one\n    return x + 1; /* really */ }\nconst char *s = "v_add // not a comment";\n'
    b = 'int f(int x) {\n\n  return x + 1; }   // reworded\nconst char *s = "v_add // not a comment";\n'
    c = a.replace("x + 1", "x + 2")
    d = a.replace("not a comment", "still not a comment")
    assert bench._code_only(a) == bench._code_only(b)
    assert bench._code_only(a) != bench._code_only(c)
    assert bench._code_only(a) != bench._code_only(d)


def test_the_concurrent_workers_leg_with_stand_in_workers():
    """bench.py --gpus N, `end_to_end_workers`: N single-stream jobs started together, every one a
    process of its own that takes a GPU through the lock slots (falcon_amd/devices.py).  Stand-in
    workers (no GPU: they take a slot among four pretended devices, say so the way the worker does,
    and print a digest of the text as their FASTA): four jobs -> four different devices, per worker a
    wall time and rates, identical outputs; a fifth doubles up."""
    import sys
    from benchlib.e2e import end_to_end_workers
    stand_in = [sys.executable, "-c", """
import hashlib, sys, time
sys.path.insert(0, %r)
from falcon_amd.devices import choose_device
d = choose_device([0, 1, 2, 3])
sys.stderr.write("INFO:falcon_amd.consensus:falcon_amd consensus on 1 engine(s), device(s) %%d (t=0.010)\\n" %% d)
text = sys.stdin.buffer.read()
time.sleep(0.4)   # (holds its slot while the others choose)
sys.stdout.write(">digest\\n%%s\\n" %% hashlib.sha1(text).hexdigest())
sys.stderr.write("INFO:falcon_amd.consensus:falcon_amd consensus: 6 piles in 3 batches; steady state 1234 piles/s (x)\\n")
""" % ROOT]
    piles = [[b"ACGT" * 300, b"ACGT" * 300, b"ACGA" * 250] for _ in range(6)]
    out = end_to_end_workers(piles, 4, repeats=2, worker_cmd=stand_in)
    assert out["distinct_devices"] == 4 and out["every_fasta_identical"], out
    assert sorted(w["devices"] for w in out["workers"]) == ["0", "1", "2", "3"]
    for w in out["workers"]:
        assert w["wall_s"] >= 0.4 and w["piles_per_sec"] > 0 and w["steady_state_piles_per_sec"] == 1234.0
    out5 = end_to_end_workers(piles, 5, repeats=1, worker_cmd=stand_in)
    assert out5["distinct_devices"] == 4 and len(out5["workers"]) == 5


def test_plumbing_under_nccl_binds_every_rank_to_its_own_device(monkeypatch):
    """What stays RCCL-only on the GPU box, checked here against a recording stand-in for torch.distributed
    and torch.cuda: for ranks 0..7 of an 8-rank node Plumbing("nccl") selects cuda:local_rank, hands
    device_id=cuda:local_rank to init_process_group (RCCL binds its communicator to that device: no lazy
    init on device 0 by every rank), verifies the world size, and bench_rank opens Engine(local_rank)."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    calls = []
    monkeypatch.delenv("FALCON_BENCH_BACKEND", raising=False)
    monkeypatch.delenv("FALCON_BENCH_ONE_DEVICE", raising=False)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append(("set_device", d)))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: calls.append(("synchronize",)))
    monkeypatch.setattr(dist, "init_process_group", lambda backend, **kw: calls.append(("init", backend, kw)))
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 8)
    monkeypatch.setattr(dist, "barrier", lambda *a, **k: calls.append(("barrier",)))
    monkeypatch.setattr(dist, "destroy_process_group", lambda *a: calls.append(("destroy",)))
    for local_rank in range(8):
        del calls[:]
        pl = bench.Plumbing(local_rank, local_rank, 8, "nccl")
        assert pl.cuda and pl.gpu and pl.device == local_rank
        assert calls[0] == ("set_device", local_rank)
        kind, backend, kw = calls[1]
        assert (kind, backend) == ("init", "nccl") and kw["rank"] == local_rank and kw["world_size"] == 8
        assert kw["device_id"] == torch.device("cuda", local_rank)
        pl.sync()
        assert calls[-2:] == [("synchronize",), ("barrier",)]
        pl.close()
        assert calls[-1] == ("destroy",)
        # the engine of the rank is opened on the same device
        opened = []

        class _E(StandInEngine):
            def __init__(self, device):
                opened.append(device)
                raise RuntimeError("stop here")
        with pytest.raises(RuntimeError, match="stop here"):
            bench.bench_rank(bench.parse_args(["--gpus", "8", "--piles", "2"]), pl, _E, [[b"ACGT"] * 3] * 2)
        assert opened == [local_rank]
    # a rendezvous that saw another number of ranks is refused
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 7)
    with pytest.raises(RuntimeError, match="rendezvous saw 7 ranks"):
        bench.Plumbing(0, 0, 8, "nccl")


def test_the_long_end_to_end_leg_with_stand_ins(tmp_path):
    """benchlib.e2e.end_to_end_long (bench.py --workload e2e-long): jobs of the whole text through one server per
    setting, a few at a time, the clock from the first job's start to the last job's end, every FASTA the same
    bytes.  Stand-ins for the server (prints its ready line, waits for SIGTERM) and for the jobs (copy stdin to
    stdout): the orchestration is what runs here, the GPU box runs it with the real ones."""
    import sys
    from benchlib.e2e import end_to_end_long
    server = [sys.executable, "-c",
              "import sys, time, signal\n"
              "signal.signal(signal.SIGTERM, lambda *a: sys.exit(0))\n"
              "print('stand-in ready'); sys.stdout.flush()\n"
              "time.sleep(600)\n"]
    job = [sys.executable, "-c", "import sys, time; time.sleep(0.05); sys.stdout.write(sys.stdin.read())"]
    piles = [[b"ACGTACGTAC" * 30, b"ACGTACGTAC" * 30, b"ACGTACGTAC" * 20] for _ in range(5)]
    res = end_to_end_long(piles, jobs=5, parallel=2, repeats=2, settings=[("a", {}), ("b", {"X": "1"})],
                          server_cmd=server, job_cmd=job)
    assert [r["setting"] for r in res["settings"]] == ["a", "b"]
    for r in res["settings"]:
        assert r["piles"] == 5 * 2 * 5 and r["every_fasta_identical"] and len(r["job_wall_s"]) == 5
        assert r["piles_per_sec"] > 0 and r["wall_s"] >= 0.05 * 3   # (five jobs, two at a time: three rounds)
