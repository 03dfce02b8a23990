"""Devices shared by batches: the replacement of the reference's process pool over piles
(falcon_kit/mains/consensus.py:264-274, falcon_kit/multiproc.py:28-36).

One engine (context) per GPU -- or several, FALCON_AMD_ENGINES_PER_DEVICE -- each with its own
work queue: a batch of piles goes to the engine with the least work queued (work per pile
varies 3x, so shards are never cut ahead of time), at most ``MAX_QUEUED`` batches resident
per engine.  Piles never interact, so there is no collective anywhere; results are put
back in input order by whoever prints them.

A batch runs in two halves (falcon_amd.h: fa_batch_submit / fa_batch_wait): submit queues
its index, chaining and alignment kernels under the engine's front lock and returns; its
consensus stage runs on a stream of its own, beside the kernels of the batches submitted
behind it -- three batches in flight per engine."""
from __future__ import annotations

import os
import threading

KMER = 8  # consensus.py:270 hard-wires K = 8


class Device:
    """One engine shared by whoever has batches for it: staging is serialised inside the
    library (the context's staging buffers), the throughput stages by ``run_lock``."""

    def __init__(self, engine, index):
        self.engine, self.index = engine, index
        self.run_lock = threading.Lock()
        self.queued = 0      # batches staged or being staged, not finished yet
        self.batches = 0     # batches this device has been given (statistics, tests)


class DevicePool:
    # batches staged (resident in HBM, ~5 GB each at the default size) but not finished, per
    # device: enough to keep the device busy, bounded however many producers there are
    MAX_QUEUED = 5

    def __init__(self, engines, grow=None):
        """grow: optional callable -> a further engine or None; asked when every device has
        MAX_QUEUED batches waiting (the single-stream worker starts on one GPU and takes another
        one only when that one is the bottleneck AND the other one is nobody's)."""
        self.devices = [Device(e, i) for i, e in enumerate(engines)]
        self._lock = threading.Condition()
        self._next = 0
        self._grow = grow
        self._growing = False

    def take(self) -> Device:
        """The device with the least work queued (round robin among equals); waits while
        every device has MAX_QUEUED batches waiting.  A further device is opened OUTSIDE the
        pool's lock (a HIP context, 104 MB of pinned memory, the code objects: a few hundred
        milliseconds during which the runners must be able to give their devices back)."""
        while True:
            grow = None
            with self._lock:
                while True:
                    n = len(self.devices)
                    order = [self.devices[(self._next + i) % n] for i in range(n)]
                    dev = min(order, key=lambda d: d.queued)
                    if dev.queued < self.MAX_QUEUED:
                        self._next = (dev.index + 1) % len(self.devices)
                        dev.queued += 1
                        dev.batches += 1
                        return dev
                    if self._grow is not None and not self._growing:
                        self._growing = True
                        grow = self._grow
                        break
                    self._lock.wait()
            engine, asked = None, False
            try:
                engine = grow()
                asked = True
            finally:
                # (a grow() that raised -- no memory for another arena -- must not leave the pool waiting
                # for a device that never comes: the flag goes back and the waiters look again)
                with self._lock:
                    self._growing = False
                    if engine is not None:
                        self.devices.append(Device(engine, len(self.devices)))
                    elif asked:
                        self._grow = None  # (nothing left to take: do not ask again at every batch)
                    self._lock.notify_all()

    def give_back(self, dev: Device):
        with self._lock:
            dev.queued -= 1
            self._lock.notify_all()

    def close(self):
        for d in self.devices:
            d.engine.close()


# ---- which GPU does a job take ------------------------------------------------------------
# fc_run starts one consensus job per LA4Falcon block, several at a time
# (falcon_kit/mains/consensus_split.py:55-85, run1.py:450), and the reference's worker sizes
# itself from --n-core alone (consensus.py:258-264).  A job whose single ingest thread cannot
# feed even one MI355X must not open a context -- arena, streams, planner thread -- on every
# GPU of the node: by default it takes ONE, chosen so that jobs started together spread over
# the node.  Advisory locks, no HIP call (looking at a device's memory would create the very
# context this is about): slot k of device d is the file falcon_amd.dev<d>.slot<k>; a job
# takes the first free (d, k) in the order k = 0, 1, ... and, inside a k, d from pid mod n on.
# Eight jobs on eight GPUs hold eight different slot-0 locks; the ninth gets slot 1 somewhere.
# The lock lives as long as the process (flock: the kernel drops it with the descriptor).
_held_locks = []
held_slots = []   # (device, slot) of every lock this process holds: what the worker's log line says


def _lock_dir():
    """Where the slot files live: one directory for every user of the node (01777, files 0666: the
    second user's job must be able to open the first user's slot files), or None when it cannot be
    had -- advisory locking must never be what fails a job."""
    d = os.environ.get("FALCON_AMD_LOCK_DIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "falcon_amd.locks")
    try:
        if not os.path.isdir(d):
            old = os.umask(0)
            try:
                os.makedirs(d, mode=0o1777, exist_ok=True)
            finally:
                os.umask(old)
    except OSError:
        return None
    return d


def _try_lock(dev, slot):
    """True: this process holds slot `slot` of device `dev` from now on.  False: somebody else does.
    None: the slot cannot be locked at all (no directory, a file of another user's that is not
    ours to open, a file system without flock) -- the caller falls back on pid mod n."""
    import fcntl
    d = _lock_dir()
    if d is None:
        return None
    path = os.path.join(d, "falcon_amd.dev%d.slot%d" % (dev, slot))
    old = os.umask(0)
    try:
        fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o666)
    except OSError:
        return None
    finally:
        os.umask(old)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
    except BlockingIOError:
        os.close(fd)
        return False
    except OSError:
        os.close(fd)
        return None
    _held_locks.append(fd)
    held_slots.append((dev, slot))
    return True


def choose_device(devices, max_slots=64, idle_only=False, skip=()):
    """One of `devices` (device ids): the first whose slot is free, slots in the order above.
    idle_only: only a device nobody holds (slot 0) -- None if there is none.  When the slots
    cannot be locked at all (_try_lock: None) the jobs spread by pid alone."""
    n = len(devices)
    first = os.getpid() % n
    order = [devices[(first + i) % n] for i in range(n) if devices[(first + i) % n] not in skip]
    for slot in range(1 if idle_only else max_slots):
        for d in order:
            got = _try_lock(d, slot)
            if got:
                return d
            if got is None:
                return None if idle_only or not order else order[0]
    return None if idle_only or not order else order[0]


def visible_devices():
    from falcon_amd.lib import load
    n_dev = load().fa_device_count()
    if n_dev <= 0:
        raise RuntimeError("falcon_amd: no HIP device visible (there is no CPU fallback)")
    return list(range(n_dev))


def open_engines(all_devices=False):
    """The engines of a worker.  FALCON_AMD_DEVICES=<ids> names them (one engine per id, times
    FALCON_AMD_ENGINES_PER_DEVICE); FALCON_AMD_DEVICES=all or all_devices=True (the
    multi-stream worker, which has the streams to feed them) takes every visible GPU; otherwise
    ONE GPU, chosen so that jobs started together spread over the node (choose_device).
    Raises without a HIP device: there is no CPU fallback."""
    from falcon_amd.engine import Engine
    devices = visible_devices()
    env = os.environ.get("FALCON_AMD_DEVICES")
    if env and env != "all":
        devices = [int(x) for x in env.split(",")]
    elif not (all_devices or env == "all"):
        devices = [choose_device(devices)]
    per_gpu = max(1, int(os.environ.get("FALCON_AMD_ENGINES_PER_DEVICE", "1")))
    return [_warmed(Engine(d)) for d in devices for _ in range(per_gpu)]


def _warmed(engine):
    """(the workers open their engines while the first batches are read: the pinned staging buffer,
    the kernels' code objects and the first upload are paid for there, not by the first batch)"""
    if hasattr(engine, "warm"):
        engine.warm(int(os.environ.get("FALCON_AMD_BATCH_BASES", "400000000")))
    return engine


def open_pool():
    """The single-stream worker's devices: one GPU (open_engines), and a way to a second one --
    only an idle one, only when the first one's queue stays full (DevicePool.take)."""
    engines = open_engines()
    grow = None
    if not os.environ.get("FALCON_AMD_DEVICES"):
        from falcon_amd.engine import Engine
        mine = [e.device for e in engines]

        def grow():
            d = choose_device(visible_devices(), idle_only=True, skip=mine)
            if d is None:
                return None
            mine.append(d)
            return _warmed(Engine(d))
    return DevicePool(engines, grow)


class EngineBackend:
    """What a worker does with a device (replaced by a stand-in in the CPU tests)."""

    def __init__(self, min_cov, min_idt):
        self.min_cov, self.min_idt = min_cov, min_idt

    def stage(self, engine, ps):
        """ps: a PileSet of the native reader, or a list of piles (lists of sequences)."""
        from falcon_amd.engine import Batch
        if isinstance(ps, list):
            return engine.batch(ps)
        return Batch.from_pileset(engine, ps)

    def submit(self, batch):
        """Throughput stages (under the device's run_lock)."""
        batch.submit(self.min_cov, KMER, self.min_idt)

    def collect(self, batch):
        """Sequential stages + download (outside the lock); frees the batch."""
        try:
            batch.wait()
            batch.fetch(False)
            return [batch.result(p) for p in range(batch.n_pile)]
        finally:
            batch.free()

    def collect_fasta(self, batch, seed_ids, mode):
        """The same, straight to the FASTA text of the batch (native output rules):
        (bytes, [(pile, reason)] of failed piles); frees the batch."""
        try:
            batch.wait()
            batch.fetch(False)
            return batch.fasta(seed_ids, mode), batch.failures()
        finally:
            batch.free()

    def release(self, batch):
        batch.free()


class SharedGpu:
    """The ``gpu`` object ``consensus._run_native`` drives: every batch goes to one device
    of the (possibly shared) pool."""

    def __init__(self, pool: DevicePool, backend):
        self.pool, self.backend = pool, backend
        self.engines = tuple(d.engine for d in pool.devices)
        # batches this worker may have between submit and collect, per engine: one in the
        # throughput stages (they take turns: the device's run_lock), two in the sequential
        # stages and the download (FALCON_AMD_RUNNERS_PER_ENGINE)
        self.parallel = max(1, int(os.environ.get("FALCON_AMD_RUNNERS_PER_ENGINE", "3"))) * len(pool.devices)

    def stage(self, ps):
        dev = self.pool.take()
        try:
            return Staged(self, dev, self.backend.stage(dev.engine, ps))
        except BaseException:
            self.pool.give_back(dev)
            raise

    def finish(self, st, fasta=None):
        """Consensus strings of a staged batch -- or, with ``fasta=(seed_ids, mode)`` and a
        backend that formats natively, (FASTA bytes, failed piles)."""
        try:
            try:
                with st.dev.run_lock:
                    st.done = True
                    self.backend.submit(st.batch)
            except BaseException:
                self.backend.release(st.batch)
                raise
            if fasta is not None:
                return self.backend.collect_fasta(st.batch, *fasta)
            return self.backend.collect(st.batch)
        finally:
            self.pool.give_back(st.dev)

    @property
    def native_fasta(self):
        return hasattr(self.backend, "collect_fasta")


class Staged:
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
