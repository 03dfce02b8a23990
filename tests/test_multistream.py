"""The multi-stream worker (falcon_amd/mains/consensus_multi.py): several pile
streams share the devices, every job's FASTA equals what the single-stream worker prints for
its stream.  CPU tests drive it with a stand-in for the engine; the gpu test runs the real
one against falcon_amd.mains.consensus."""
import io
import os
import random
import threading
import time

import pytest

from falcon_amd.mains import consensus as single
from falcon_amd.mains import consensus_multi as multi
from test_native_reader import _rand_stream

OPTS = ["--min-n-read", "2", "--min-cov-aln", "0", "--output-full"]


class FakeEngine:
    def close(self):
        pass


class FakeBackend:
    """Consensus of a pile := 600 characters of its seed, repeated (>= 500: printed).
    ``submit`` = the throughput stages (one batch at a time per engine: the engine's lock),
    ``collect`` = the sequential stages + download (beside the next batch's ``submit``)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.staged = self.finished = self.released = 0
        self.running = {}       # engine -> batches in submit() right now (must never exceed 1)
        self.overlap = False
        self.in_flight = {}     # engine -> batches between submit and the end of collect
        self.peak_in_flight = 0

    def stage(self, engine, ps):
        with self.lock:
            self.staged += 1
        return (engine, ps.piles())

    def submit(self, batch):
        engine, piles = batch
        with self.lock:
            self.running[engine] = self.running.get(engine, 0) + 1
            self.overlap |= self.running[engine] > 1
            self.in_flight[engine] = self.in_flight.get(engine, 0) + 1
            self.peak_in_flight = max(self.peak_in_flight, self.in_flight[engine])
        time.sleep(0.002)
        with self.lock:
            self.running[engine] -= 1

    def collect(self, batch):
        engine, piles = batch
        time.sleep(0.002)
        with self.lock:
            self.in_flight[engine] -= 1
            self.finished += 1
        return [(p[0] * 20)[:600] for p in piles]

    def release(self, batch):
        with self.lock:
            self.released += 1


def _single_stream_output(text, args):
    """What the single-stream worker prints for `text` with the same stand-in."""
    class Gpu:
        engines = [None]

        def stage(self, ps):
            return ps.piles()

        def finish(self, piles):
            return [(p[0] * 20)[:600] for p in piles]

    rd, wr = os.pipe()
    t = threading.Thread(target=lambda: (os.write(wr, text.encode()), os.close(wr)))
    t.start()
    out = io.StringIO()
    try:
        single._run_native(args, single.settings_from(args), rd, Gpu(), out, batch_bases=900)
    finally:
        os.close(rd)
        t.join()
    return out.getvalue()


def test_batches_that_grow_print_the_same_bytes(monkeypatch):
    """FALCON_AMD_BATCH_GROW=FROM:FACTOR:MAX: from batch FROM on every batch is FACTOR times the one before, up to
    MAX bases (for streams that go on: the GPU is faster on large batches).  The records leave in input order and
    the bytes are those of constant batches; the schedule itself is checked, and what it refuses."""
    f = single.batch_schedule(900, "2:2:5000")
    assert [f(n) for n in range(7)] == [900, 900, 1800, 3600, 5000, 5000, 5000]
    assert [single.batch_schedule(900, None)(n) for n in (0, 5, 500)] == [900] * 3
    assert single.batch_schedule(400_000_000, "3:1.5:1600000000")(200) == 1_600_000_000
    for bad in ("2:0.5:5000", "-1:2:5000", "2:2:100"):
        with pytest.raises(ValueError):
            single.batch_schedule(900, bad)
    rng = random.Random(35)
    text = _rand_stream(rng, 150, with_noise=True)
    args = single.parse_args(["prog"] + OPTS)
    want = _single_stream_output(text, args)
    sizes = []

    class Gpu:
        engines = [None]
        parallel = 3

        def stage(self, ps):
            sizes.append(ps.n_pile)
            return ps.piles()

        def finish(self, piles):
            return [(p[0] * 20)[:600] for p in piles]

    monkeypatch.setenv("FALCON_AMD_BATCH_GROW", "3:2:20000")
    rd, wr = os.pipe()
    t = threading.Thread(target=lambda: (os.write(wr, text.encode()), os.close(wr)))
    t.start()
    out = io.StringIO()
    try:
        single._run_native(args, single.settings_from(args), rd, Gpu(), out, batch_bases=900)
    finally:
        os.close(rd)
        t.join()
    assert out.getvalue() == want and want.count(">") > 80
    assert max(sizes[5:]) > 4 * max(sizes[:3]), sizes   # (the later batches ARE larger)


@pytest.mark.parametrize("stagers", ["2", "4"])
def test_several_staging_threads_keep_the_order_and_the_readers_buffers(stagers, monkeypatch):
    """FALCON_AMD_STAGERS > 1: batches are staged side by side and may finish out of order -- the records
    still leave in stream order, and a batch's text is still there while it is being staged however far
    the threads beside it have got (the reader's buffers go back in stream order; it keeps two here, so a
    buffer given back early would be overwritten by the next call)."""
    monkeypatch.setenv("FALCON_AMD_READ_AHEAD", "2")
    rng = random.Random(33)
    text = _rand_stream(rng, 150, with_noise=True)
    args = single.parse_args(["prog"] + OPTS)
    want = _single_stream_output(text, args)
    assert want.count(">") > 80
    monkeypatch.setenv("FALCON_AMD_STAGERS", stagers)
    order = []

    class Gpu:
        engines = [None]
        parallel = 3

        def stage(self, ps):
            # every third batch takes its time BEFORE it looks at its piles
            n = len(order)
            order.append(n)
            if n % 3 == 0:
                time.sleep(0.02)
            return ps.piles()

        def finish(self, piles):
            return [(p[0] * 20)[:600] for p in piles]

    rd, wr = os.pipe()
    t = threading.Thread(target=lambda: (os.write(wr, text.encode()), os.close(wr)))
    t.start()
    out = io.StringIO()
    try:
        single._run_native(args, single.settings_from(args), rd, Gpu(), out, batch_bases=900)
    finally:
        os.close(rd)
        t.join()
    assert len(order) > 20 and out.getvalue() == want


def test_jobs_share_devices_and_print_what_the_single_worker_prints(tmp_path, monkeypatch):
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    rng = random.Random(9)
    texts = [_rand_stream(rng, rng.randint(15, 40), with_noise=i % 2 == 1) for i in range(5)]
    argv = ["prog"] + OPTS
    for i, text in enumerate(texts):
        path = tmp_path / ("piles_%d.txt" % i)
        path.write_text(text)
        src = str(path) if i != 2 else "cmd:cat %s" % path   # one job reads a producer's pipe
        argv += ["--job", src, str(tmp_path / ("cns_%d.fasta" % i))]
    args = multi.parse_args(argv)
    backend = FakeBackend()
    pool = multi.DevicePool([FakeEngine(), FakeEngine(), FakeEngine()])
    results = multi.run(args, pool=pool, backend=backend)
    assert [r[2] for r in results] == [None] * len(texts)
    for i, text in enumerate(texts):
        got = (tmp_path / ("cns_%d.fasta" % i)).read_text()
        assert got == _single_stream_output(text, args), i
        assert not (tmp_path / ("cns_%d.fasta.tmp" % i)).exists()
    assert backend.staged == backend.finished > len(texts) and backend.released == 0
    assert not backend.overlap                       # one batch at a time in a device's throughput stages
    assert all(d.batches > 0 and d.queued == 0 for d in pool.devices)


def test_single_stream_worker_on_several_engines(tmp_path):
    """falcon_amd.mains.consensus on a node with several GPUs: GpuConsensus with
    three stand-in engines behind the worker's pipeline -- batches go to the engine with the
    least work queued (no contiguous split ahead of time, no per-batch join), three batches per
    engine in flight (one in the throughput stages), records printed in input order; the
    python-parser path (``imap``, e.g. --trim) shares the same queues."""
    rng = random.Random(21)
    text = _rand_stream(rng, 120, with_noise=True)
    args = single.parse_args(["prog"] + OPTS)
    cfg = single.settings_from(args)
    want = _single_stream_output(text, args)
    assert want.count(">") > 60

    class SlowFirst(FakeBackend):
        """engine 0 is slow: a static split would wait for it, the queues route around it"""
        def submit(self, batch):
            if batch[0].name == "dev0":
                time.sleep(0.03)
            super().submit(batch)

    class Named(FakeEngine):
        def __init__(self, name):
            self.name, self.closed = name, False

        def close(self):
            self.closed = True

    engines = [Named("dev%d" % i) for i in range(3)]
    backend = SlowFirst()
    gpu = single.GpuConsensus(args.min_cov, args.min_idt, engines=engines, backend=backend)
    assert gpu.parallel == 9 and len(gpu.engines) == 3
    rd, wr = os.pipe()
    t = threading.Thread(target=lambda: (os.write(wr, text.encode()), os.close(wr)))
    t.start()
    out = io.StringIO()
    try:
        single._run_native(args, cfg, rd, gpu, out, batch_bases=900)
    finally:
        os.close(rd)
        t.join()
    assert out.getvalue() == want                     # input order, nothing lost
    per_dev = [d.batches for d in gpu.pool.devices]
    assert all(n > 0 for n in per_dev) and per_dev[0] < min(per_dev[1:])   # least loaded first
    assert not backend.overlap and 2 <= backend.peak_in_flight <= multi.DevicePool.MAX_QUEUED
    assert backend.staged == backend.finished and all(d.queued == 0 for d in gpu.pool.devices)
    # the python-parser path over the same pool
    piles = [pile for _, pile in single.PileReader(io.StringIO(text), cfg, args.min_n_read, args.min_len_aln)]
    gpu.batch_bases = 900

    class ListBackend(SlowFirst):
        def stage(self, engine, ps):
            with self.lock:
                self.staged += 1
            return (engine, ps)
    gpu.shared.backend = ListBackend()
    got = list(gpu.imap(iter(piles)))
    assert got == [(p[0] * 20)[:600] for p in piles]
    gpu.close()
    assert all(e.closed for e in engines)


def test_more_jobs_than_queue_slots_still_finish(tmp_path, monkeypatch):
    """Twelve jobs on one device with at most two batches queued: nobody starves or deadlocks."""
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    monkeypatch.setattr(multi.DevicePool, "MAX_QUEUED", 2)
    rng = random.Random(11)
    argv, texts = ["prog"] + OPTS, []
    for i in range(12):
        texts.append(_rand_stream(rng, rng.randint(5, 25), with_noise=False))
        (tmp_path / ("p%d.txt" % i)).write_text(texts[-1])
        argv += ["--job", str(tmp_path / ("p%d.txt" % i)), str(tmp_path / ("c%d.fasta" % i))]
    args = multi.parse_args(argv)
    backend = FakeBackend()
    pool = multi.DevicePool([FakeEngine()])
    peak = []
    take = pool.take

    def watched():
        d = take()
        peak.append(d.queued)
        return d
    pool.take = watched
    res = multi.run(args, pool=pool, backend=backend)
    assert [r[2] for r in res] == [None] * 12 and max(peak) <= 2 and not backend.overlap
    for i, text in enumerate(texts):
        assert (tmp_path / ("c%d.fasta" % i)).read_text() == _single_stream_output(text, args)


def test_a_failing_job_is_reported_and_the_others_finish(tmp_path, monkeypatch):
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    rng = random.Random(10)
    text = _rand_stream(rng, 20, with_noise=False)
    good = tmp_path / "good.txt"
    good.write_text(text)

    class Flaky(FakeBackend):
        def collect(self, batch):
            if any(p[0].startswith("BOOM") for p in batch[1]):
                raise RuntimeError("device fault")
            return super().collect(batch)

    bad = tmp_path / "bad.txt"
    bad.write_text("s BOOM%s\nr1 ACGTACGTACGT\nr2 ACGTACGTACG\n+ +\n" % ("A" * 40) + text)
    args = multi.parse_args(["prog"] + OPTS + [
        "--job", str(good), str(tmp_path / "good.fasta"),
        "--job", str(bad), str(tmp_path / "bad.fasta"),
        "--job", str(tmp_path / "missing.txt"), str(tmp_path / "missing.fasta"),
        "--job", "cmd:exit 3", str(tmp_path / "producer_failed.fasta")])
    backend = Flaky()
    pool = multi.DevicePool([FakeEngine(), FakeEngine()])
    res = multi.run(args, pool=pool, backend=backend)
    assert res[0][2] is None and (tmp_path / "good.fasta").read_text() == _single_stream_output(text, args)
    assert isinstance(res[1][2], RuntimeError) and "device fault" in str(res[1][2])
    assert isinstance(res[2][2], FileNotFoundError)
    assert isinstance(res[3][2], RuntimeError) and "status 3" in str(res[3][2])
    for name in ("bad", "missing", "producer_failed"):
        assert not (tmp_path / (name + ".fasta")).exists()
        assert not (tmp_path / (name + ".fasta.tmp")).exists()
    assert all(d.queued == 0 for d in pool.devices)   # nothing left queued on a device


def test_command_line_rules():
    with pytest.raises(SystemExit):
        multi.parse_args(["prog"] + OPTS)                                   # no job
    with pytest.raises(SystemExit):
        multi.parse_args(["prog", "--trim", "--job", "a", "b"])             # --trim: single worker
    with pytest.raises(SystemExit):
        multi.parse_args(["prog", "--job", "a", "x", "--job", "b", "x"])    # same output twice
    with pytest.raises(SystemExit):
        multi.parse_args(["prog", "--job", "-", "x", "--job", "-", "y"])    # stdin twice
    with pytest.raises(SystemExit):
        multi.parse_args(["prog", "--job", "a"])                            # incomplete
    a = multi.parse_args(["prog", "--min-cov", "4", "--job", "a", "x", "--output-multi", "--job", "b", "y"])
    assert a.jobs == [("a", "x"), ("b", "y")] and a.min_cov == 4 and a.output_multi


@pytest.mark.gpu
def test_multi_stream_worker_on_the_gpu(tmp_path):
    """Two jobs on the real engine: each FASTA equals the single-stream worker's stdout."""
    from falcon_amd.synth import make_pile, pile_to_la4falcon
    opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
    argv = ["prog"] + opts
    texts = []
    for j in range(2):
        chunks = []
        for s in range(3):
            seed, rd = make_pile(900 + 10 * j + s, S=2500, coverage=14, min_read=500, mean_read=1500,
                                 sd_read=400)
            chunks.append(pile_to_la4falcon("%09d" % (10 * j + s), seed, rd, 1000 * (10 * j + s) + 1))
        text = "".join(chunks) + "- -\n"
        path = tmp_path / ("piles_%d.txt" % j)
        path.write_text(text)
        texts.append(path)
        argv += ["--job", str(path), str(tmp_path / ("cns_%d.fasta" % j))]
    res = multi.run(multi.parse_args(argv))
    assert [r[2] for r in res] == [None, None]
    for j, path in enumerate(texts):
        out = io.StringIO()
        single.run(single.parse_args(["prog"] + opts), stdin=open(path), stdout=out)
        got = (tmp_path / ("cns_%d.fasta" % j)).read_text()
        assert got == out.getvalue() and got.count(">") >= 3


def test_a_job_with_an_uncorrected_pile_fails_alone(tmp_path, monkeypatch):
    """Piles fail alone (fa_batch_pile_error); the multi-stream worker must not deliver a
    FASTA that silently lacks them: that job is reported failed and leaves no output file,
    the other jobs are untouched, FALCON_AMD_SKIP_FAILED_PILES=1 accepts the gap (the
    single-stream worker's exit status 3 and its opt-out)."""
    from falcon_amd.engine import FailedPile
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    rng = random.Random(5)
    texts = [_rand_stream(rng, 12, with_noise=False) for _ in range(3)]
    marker = texts[1].split("\n")[0].split()[0]  # first seed of job 1: the pile that "fails"

    class Backend(FakeBackend):
        def stage(self, engine, ps):
            with self.lock:
                self.staged += 1
            return (engine, ps.piles(), list(ps.seed_ids))

        def submit(self, batch):
            pass

        def collect(self, batch):
            engine, piles, ids = batch
            return [FailedPile("too deep") if i == marker else (p[0] * 20)[:600] for i, p in zip(ids, piles)]

    def go():
        argv = ["prog"] + OPTS
        for i, text in enumerate(texts):
            path = tmp_path / ("in_%d.txt" % i)
            path.write_text(text)
            argv += ["--job", str(path), str(tmp_path / ("out_%d.fasta" % i))]
        args = multi.parse_args(argv)
        return multi.run(args, pool=multi.DevicePool([FakeEngine()]), backend=Backend())

    res = go()
    assert res[0][2] is None and res[2][2] is None
    assert isinstance(res[1][2], RuntimeError) and marker in str(res[1][2])
    assert (tmp_path / "out_0.fasta").exists() and (tmp_path / "out_2.fasta").exists()
    assert not (tmp_path / "out_1.fasta").exists() and not (tmp_path / "out_1.fasta.tmp").exists()
    monkeypatch.setenv("FALCON_AMD_SKIP_FAILED_PILES", "1")
    res = go()
    assert [r[2] for r in res] == [None] * 3
    assert marker not in (tmp_path / "out_1.fasta").read_text()


class TimedBackend(FakeBackend):
    """A device modelled in time: staging a batch 8 ms of host work (copy into pinned memory,
    upload), its throughput stages 20 ms under the engine's lock, the sequential stages and
    the download 8 ms beside the next batch's."""

    def stage(self, engine, ps):
        time.sleep(0.008)
        return super().stage(engine, ps)

    def submit(self, batch):
        engine, piles = batch
        with self.lock:
            self.running[engine] = self.running.get(engine, 0) + 1
            self.overlap |= self.running[engine] > 1
        time.sleep(0.020)
        with self.lock:
            self.running[engine] -= 1

    def collect(self, batch):
        engine, piles = batch
        time.sleep(0.008)
        with self.lock:
            self.finished += 1
        return [(p[0] * 20)[:600] for p in piles]


def test_four_devices_are_fed_by_four_streams(tmp_path, monkeypatch):
    """piles shard over the GPUs of a node with no collective -- but somebody has
    to FEED them.  One stream has one reader and one staging thread, which a single device's
    20 ms batches already keep busy; N devices are fed by N streams through the
    multi-stream worker, each with its own reader and stager (one consensus job per .las
    block is what fc_run starts anyway).  With a device modelled in time, four streams on four
    engines must take at most a third of what they take on one -- a single staging thread, or
    a lock shared by the devices, would show here -- and no engine ever runs two batches'
    throughput stages at once."""
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    rng = random.Random(12)
    texts = [_rand_stream(rng, 60, with_noise=False) for _ in range(4)]

    def run_on(n_engines, tag):
        argv = ["prog"] + OPTS
        for i, text in enumerate(texts):
            path = tmp_path / ("s_%s_%d.txt" % (tag, i))
            path.write_text(text)
            argv += ["--job", str(path), str(tmp_path / ("o_%s_%d.fasta" % (tag, i)))]
        args = multi.parse_args(argv)
        backend = TimedBackend()
        pool = multi.DevicePool([FakeEngine() for _ in range(n_engines)])
        t0 = time.perf_counter()
        results = multi.run(args, pool=pool, backend=backend)
        wall = time.perf_counter() - t0
        assert [r[2] for r in results] == [None] * len(texts)
        assert not backend.overlap
        return wall, backend.finished, [(tmp_path / ("o_%s_%d.fasta" % (tag, i))).read_text() for i in range(4)]

    wall1, n1, out1 = run_on(1, "one")
    wall4, n4, out4 = run_on(4, "four")
    assert out1 == out4 and n1 == n4 and n1 >= 40
    assert wall1 >= n1 * 0.020            # one engine: its throughput stages in a row
    assert wall4 <= wall1 / 3.0, (wall1, wall4, n1)


def test_jobs_hand_their_descriptors_to_a_server(tmp_path, monkeypatch):
    """falcon_amd/mains/consensus_server.py with stand-in devices: several job PROCESSES (the console
    command with FALCON_AMD_SERVER set) hand their stdin / stdout to one server over SCM_RIGHTS, some at
    the same time; each gets the bytes the stand-alone worker prints for its stream and status 0; a
    --trim job is declined and runs by itself (here: fails for want of a GPU, which proves it did);
    without a server behind the socket a job also runs by itself."""
    import subprocess
    import sys
    from falcon_amd.mains import consensus_server as server
    monkeypatch.setenv("FALCON_AMD_BATCH_BASES", "900")
    rng = random.Random(33)
    texts = [_rand_stream(rng, rng.randint(15, 40), with_noise=i % 2 == 1) for i in range(4)]
    sock = str(tmp_path / "srv.sock")
    backend = FakeBackend()
    pool = multi.DevicePool([FakeEngine(), FakeEngine()])
    stop = threading.Event()
    th = threading.Thread(target=server.serve, args=(sock,), kwargs=dict(pool=pool, make_backend=lambda a: backend, stop=stop))
    th.start()
    try:
        for _ in range(200):
            if os.path.exists(sock):
                break
            time.sleep(0.01)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, FALCON_AMD_SERVER=sock, PYTHONPATH=root)
        procs = []
        for i, text in enumerate(texts):
            (tmp_path / ("in_%d.txt" % i)).write_text(text)
            fin, fout = open(tmp_path / ("in_%d.txt" % i)), open(tmp_path / ("out_%d.fa" % i), "w")
            procs.append((subprocess.Popen([sys.executable, "-m", "falcon_amd.mains.consensus"] + OPTS, stdin=fin, stdout=fout,
                                           stderr=subprocess.PIPE, env=env, cwd=root), fin, fout))
        args = single.parse_args(["prog"] + OPTS)
        for i, (p, fin, fout) in enumerate(procs):
            _, err = p.communicate(timeout=120)
            fin.close(); fout.close()
            assert p.returncode == 0, err.decode()
            assert (tmp_path / ("out_%d.fa" % i)).read_text() == _single_stream_output(texts[i], args), i
        assert backend.staged == backend.finished > len(texts)
        # a second server on the same path does not take the socket from a live one
        with pytest.raises(RuntimeError, match="already listening"):
            server.serve(sock, pool=multi.DevicePool([FakeEngine()]), make_backend=lambda a: backend)
        assert os.path.exists(sock)
        # a job the server declines runs in its own process: no GPU here, so it fails -- by itself
        with open(tmp_path / "in_0.txt") as fin:
            p = subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus", "--trim", "--n-core", "1"] + OPTS, stdin=fin,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=root, timeout=120)
        assert p.returncode != 0 and b"HIP" in p.stderr + p.stdout
    finally:
        stop.set()
        th.join(timeout=30)
    assert not os.path.exists(sock)
    # nobody behind the variable: the job goes on by itself (and, here, fails for want of a GPU)
    with open(tmp_path / "in_0.txt") as fin:
        p = subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus", "--n-core", "1"] + OPTS, stdin=fin,
                           stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, FALCON_AMD_SERVER=sock, PYTHONPATH=root),
                           cwd=root, timeout=120)
    assert p.returncode != 0 and b"HIP" in p.stderr + p.stdout
