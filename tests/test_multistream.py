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
    """Consensus of a pile := 600 characters of its seed, repeated (>= 500: printed)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.staged = self.finished = self.released = 0
        self.running = {}       # engine -> batches in finish() right now (must never exceed 1)
        self.overlap = False

    def stage(self, engine, ps):
        with self.lock:
            self.staged += 1
        return (engine, ps.piles())

    def finish(self, batch):
        engine, piles = batch
        with self.lock:
            self.running[engine] = self.running.get(engine, 0) + 1
            self.overlap |= self.running[engine] > 1
        time.sleep(0.002)
        with self.lock:
            self.running[engine] -= 1
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
    assert not backend.overlap                       # one batch at a time per device
    assert all(d.batches > 0 and d.queued == 0 for d in pool.devices)


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
        def finish(self, batch):
            if any(p[0].startswith("BOOM") for p in batch[1]):
                raise RuntimeError("device fault")
            return super().finish(batch)

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
