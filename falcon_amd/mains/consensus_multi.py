"""Several LA4Falcon pile streams through ONE worker process on the node's GPUs.

The reference runs one consensus job per ``.las`` block, each its own process reading
its own single-threaded ``LA4Falcon`` (falcon_kit/mains/consensus_task.py:70-104, the
pipe at :90).  One such producer cannot keep a GPU busy, let alone eight, so this worker
takes any number of jobs at once ("multi-stream input"):

    python -m falcon_amd.mains.consensus_multi --output-multi --min-idt 0.70 --min-cov 4 \\
        --max-n-read 200 \\
        --job 'cmd:LA4Falcon -H12000 -fo raw_reads raw_reads.1.las' cns_00001.fasta \\
        --job 'cmd:LA4Falcon -H12000 -fo raw_reads raw_reads.2.las' cns_00002.fasta \\
        --job piles_3.txt cns_00003.fasta

A job is a pile stream in (a file or FIFO path, ``cmd:<shell command>`` whose stdout is
the stream, or ``-`` for this process's stdin) and a FASTA file out.  Every job gets the
bytes ``falcon_amd.mains.consensus`` -- and so the reference worker -- prints for its
stream, written to ``<out>.tmp`` and renamed when the job is complete (the reference's
task does the same around the worker, consensus_task.py:74,102).  Jobs run concurrently:
each has its own native reader, staging and printer threads (the three-stage pipeline of
``consensus._run_native``) and shares the devices' engines, a batch going to the device
with the least work queued; piles of different jobs never interact, results keep their
job's input order.

There is no CPU fallback: without libfalcon_amd.so and a HIP device the command fails.
"""
from __future__ import annotations

import logging
import os
import subprocess
import sys
import threading

from falcon_amd.mains import consensus as single

LOG = logging.getLogger("falcon_amd.consensus_multi")


class Device:
    """One engine (one GPU) shared by the jobs: staging is serialised inside the library
    (the context's staging buffers), the GPU stages by ``run_lock``."""

    def __init__(self, engine, index):
        self.engine, self.index = engine, index
        self.run_lock = threading.Lock()
        self.queued = 0      # batches staged or being staged, not finished yet
        self.batches = 0     # batches this device has been given (statistics, tests)


class DevicePool:
    # batches staged (resident in HBM, ~5 GB each at the default size) but not finished, per
    # device: enough to keep the device busy, bounded however many jobs there are
    MAX_QUEUED = 4

    def __init__(self, engines):
        self.devices = [Device(e, i) for i, e in enumerate(engines)]
        self._lock = threading.Condition()
        self._next = 0

    def take(self) -> Device:
        """The device with the least work queued (round robin among equals); waits while
        every device has MAX_QUEUED batches waiting."""
        with self._lock:
            n = len(self.devices)
            while True:
                order = [self.devices[(self._next + i) % n] for i in range(n)]
                dev = min(order, key=lambda d: d.queued)
                if dev.queued < self.MAX_QUEUED:
                    break
                self._lock.wait()
            self._next = (dev.index + 1) % n
            dev.queued += 1
            dev.batches += 1
            return dev

    def give_back(self, dev: Device):
        with self._lock:
            dev.queued -= 1
            self._lock.notify_all()

    def close(self):
        for d in self.devices:
            d.engine.close()


class EngineBackend:
    """What a job does with a device (replaced by a stand-in in the CPU tests)."""

    def __init__(self, min_cov, min_idt):
        self.min_cov, self.min_idt = min_cov, min_idt

    def stage(self, engine, ps):
        from falcon_amd.engine import Batch
        return Batch.from_pileset(engine, ps)

    def finish(self, batch):
        try:
            batch.run(self.min_cov, single.KMER, self.min_idt).fetch(False)
            return [batch.result(p) for p in range(batch.n_pile)]
        finally:
            batch.free()

    def release(self, batch):
        batch.free()


class SharedGpu:
    """The ``gpu`` object ``consensus._run_native`` drives, for one job: every batch goes
    to one device of the shared pool."""

    engines = (None,)  # (_run_native sizes a batch per entry: one device per batch here)

    def __init__(self, pool: DevicePool, backend):
        self.pool, self.backend = pool, backend

    def stage(self, ps):
        dev = self.pool.take()
        try:
            return [_Staged(self, dev, self.backend.stage(dev.engine, ps))]
        except BaseException:
            self.pool.give_back(dev)
            raise

    def finish(self, batches):
        out = []
        for st in batches:
            try:
                with st.dev.run_lock:
                    st.done = True
                    out.extend(self.backend.finish(st.batch))
            finally:
                self.pool.give_back(st.dev)
        return out


class _Staged:
    """A staged batch and the device it sits on (``free``: dropped unfinished, on errors)."""

    def __init__(self, gpu, dev, batch):
        self.gpu, self.dev, self.batch, self.done = gpu, dev, batch, False

    def free(self):
        if not self.done:
            self.done = True
            try:
                self.gpu.backend.release(self.batch)
            finally:
                self.gpu.pool.give_back(self.dev)


def parse_args(argv):
    """The worker's options (consensus.py:219-250) plus ``--job SRC OUT`` (repeatable)."""
    jobs, rest, i = [], [argv[0]] if argv else ["consensus_multi"], 1
    while i < len(argv):
        if argv[i] == "--job":
            if i + 2 >= len(argv):
                raise SystemExit("--job takes two arguments: the pile stream and the FASTA file")
            jobs.append((argv[i + 1], argv[i + 2]))
            i += 3
        else:
            rest.append(argv[i])
            i += 1
    args = single.parse_args(rest)
    if not jobs:
        raise SystemExit("no --job SRC OUT given")
    if args.trim:
        raise SystemExit("--trim is not available in the multi-stream worker "
                         "(use falcon_amd.mains.consensus per stream)")
    if sum(1 for src, _ in jobs if src == "-") > 1:
        raise SystemExit("only one job can read this process's stdin")
    if len({out for _, out in jobs}) != len(jobs):
        raise SystemExit("two jobs write the same file")
    args.jobs = jobs
    return args


def _open_source(src):
    """(fd, close) for a job's pile stream."""
    if src == "-":
        return sys.stdin.fileno(), lambda: None
    if src.startswith("cmd:"):
        proc = subprocess.Popen(src[4:], shell=True, stdout=subprocess.PIPE)

        def close():
            proc.stdout.close()
            rc = proc.wait()
            if rc not in (0, -13, 141):  # (SIGPIPE, directly or through the shell: we stop reading at "- -")
                raise RuntimeError("producer %r exited with status %d" % (src[4:], rc))
        return proc.stdout.fileno(), close
    fd = os.open(src, os.O_RDONLY)
    return fd, lambda: os.close(fd)


def run_job(args, cfg, pool, backend, src, out_path):
    """One job: stream in, FASTA out (``<out>.tmp`` renamed on completion)."""
    fd, close = _open_source(src)
    tmp = out_path + ".tmp"
    try:
        try:
            with open(tmp, "w") as out:
                single._run_native(args, cfg, fd, SharedGpu(pool, backend), out)
        finally:
            close()  # (a producer that failed fails the job)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise
    os.rename(tmp, out_path)


def run(args, pool=None, backend=None):
    """Run all jobs; returns the list of (src, out, error-or-None)."""
    logging.basicConfig(level=int(round(10 * args.verbose_level)))
    cfg = single.settings_from(args)
    own_pool = pool is None
    if own_pool:
        from falcon_amd.engine import Engine
        from falcon_amd.lib import load
        n_dev = load().fa_device_count()
        if n_dev <= 0:
            raise RuntimeError("falcon_amd: no HIP device visible (there is no CPU fallback)")
        env = os.environ.get("FALCON_AMD_DEVICES")
        devices = [int(x) for x in env.split(",")] if env else list(range(n_dev))
        # engines (contexts, each with its own streams and work slots) per GPU: with 2, the
        # latency-bound consensus kernels of one job's batch run next to the alignment
        # kernel of another's (+6 % at 3072-pile batches with two contexts;
        # not measured at the worker's batch size yet, hence 1)
        per_gpu = max(1, int(os.environ.get("FALCON_AMD_ENGINES_PER_DEVICE", "1")))
        pool = DevicePool([Engine(d) for d in devices for _ in range(per_gpu)])
    if backend is None:
        backend = EngineBackend(args.min_cov, args.min_idt)
    LOG.info("falcon_amd consensus: %d job(s) on %d GPU(s)", len(args.jobs), len(pool.devices))
    results = [None] * len(args.jobs)

    def work(i):
        src, out = args.jobs[i]
        try:
            run_job(args, cfg, pool, backend, src, out)
            results[i] = (src, out, None)
        except BaseException as exc:  # reported per job; the other jobs go on
            LOG.error("job %s -> %s failed: %r", src, out, exc)
            results[i] = (src, out, exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(args.jobs))]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        if own_pool:
            pool.close()
    return results


def main(argv=None):
    args = parse_args(sys.argv if argv is None else argv)
    failed = [r for r in run(args) if r[2] is not None]
    for src, out, exc in failed:
        sys.stderr.write("falcon_amd: job %s -> %s failed: %s\n" % (src, out, exc))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
