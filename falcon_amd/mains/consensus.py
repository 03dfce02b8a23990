"""falcon_sense on MI355X: drop-in for ``python -m falcon_kit.mains.consensus``.

Same flags, same stdin grammar and the same stdout bytes as the reference worker
(/root/reference/falcon_kit/mains/consensus.py), so the pipe that fc_run writes,

    LA4Falcon -H$CUTOFF -fo db las | python -m falcon_kit.mains.consensus <opts> > cns.fasta

(falcon_kit/mains/consensus_task.py:90) keeps working unchanged.  What differs is
what sits between the parser and the printer: the reference hands one pile at a
time to a pool of worker processes (consensus.py:264-274); here piles are gathered
into batches, staged in HBM once, and every stage of generate_consensus runs as HIP
kernels over the whole batch (falcon_amd/csrc).  Records are printed in input
order, like ``Pool.imap``.

There is no CPU fallback: without libfalcon_amd.so and a HIP device the command
fails.
"""
from __future__ import annotations

import argparse
import logging
import multiprocessing
import os
import re
import sys
import threading
import time
from collections import namedtuple

LOG = logging.getLogger("falcon_amd.consensus")
_T0 = time.perf_counter()  # (debug lines carry seconds since the module was loaded)
_LEFT_OPEN = []             # what run(leave_open=True) did not close: kept from the garbage collector until _leave


def _clock():
    return time.perf_counter() - _T0


KMER = 8              # consensus.py:270 hard-wires K = 8
MAX_SEQ_LEN = 100000  # consensus.py:162: longer sequences are cut to MAX_SEQ_LEN - 1

# (flag, type, default, help) -- names, types and defaults of consensus.py:219-250
_OPTIONS = (
    ("--n-core", int, 24, "worker count requested by the caller (accepted for compatibility; "
                          "the GPU path does not fork workers)"),
    ("--min-cov", int, 6, "positions covered by at most this many reads are lower-cased and "
                          "split the output"),
    ("--min-cov-aln", int, 10, "skip seeds whose pile has less than this average depth"),
    ("--max-cov-aln", int, 0, "if > 0, stop adding (longest-first) reads past this average depth"),
    ("--min-len-aln", int, 0, "ignore sequences shorter than this"),
    ("--min-n-read", int, 10, "skip seeds with fewer sequences than this (seed copy included)"),
    ("--max-n-read", int, 500, "use at most this many sequences per seed (seed included)"),
    ("--min-idt", float, 0.70, "minimum identity of an alignment used for correction"),
    ("--edge-tolerance", int, 1000, "--trim: drop reads whose unaligned ends exceed this"),
    ("--trim-size", int, 50, "--trim: bases cut from both ends of the mapped window"),
)
_SWITCHES = (
    ("--trim", "window every read to its chained k-mer span before the consensus"),
    ("--output-full", "print the raw consensus, lower-case regions included"),
    ("--output-multi", "print every well-covered region of >= 500 bp (at most 10)"),
)

Settings = namedtuple("Settings", "min_cov K max_n_read min_idt edge_tolerance trim_size "
                                  "min_cov_aln max_cov_aln")


def parse_args(argv):
    ap = argparse.ArgumentParser(
        prog=os.path.basename(argv[0]) if argv else None,
        description="pre-assembly consensus (falcon_sense) on AMD MI355X GPUs",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag, typ, default, text in _OPTIONS:
        ap.add_argument(flag, type=typ, default=default, help=text)
    for flag, text in _SWITCHES:
        ap.add_argument(flag, action="store_true", default=False, help=text)
    ap.add_argument("-v", "--verbose-level", type=float, default=2.0,
                    help="logging level = 10 * value (1 DEBUG, 2 INFO, 3 WARNING)")
    return ap.parse_args(argv[1:])


def settings_from(args) -> Settings:
    return Settings(args.min_cov, KMER, args.max_n_read, args.min_idt, args.edge_tolerance,
                    args.trim_size, args.min_cov_aln, args.max_cov_aln)


# --------------------------------------------------------------------------
# pile selection and stream parsing (behaviour of consensus.py:26-45, :161-209)
# --------------------------------------------------------------------------
def select_reads(pile, max_n_read, max_cov_aln, presorted=False):
    """Seed first, then the longest reads.

    The sort is stable on descending length (ties keep stream order).  The result
    holds at most ``max_n_read`` sequences; with ``max_cov_aln`` > 0 reads are only
    added while (bases so far) // seed_len has not exceeded it."""
    seed, rest = pile[:1], pile[1:]
    if not presorted:
        rest = sorted(rest, key=len, reverse=True)  # python's sort is stable, also reversed
    keep = max_n_read
    if max_cov_aln > 0:
        seed_len = len(seed[0])
        depth_bases = 0
        keep = 1
        for r in rest:
            if depth_bases // seed_len > max_cov_aln:
                break
            keep += 1
            depth_bases += len(r)
        keep = min(keep, max_n_read)
    return (seed + rest)[:keep]


class PileReader:
    """Iterates over the admitted piles of an LA4Falcon ``-fo`` text stream.

    Grammar: ``<id> <bases>`` lines; ``+ +`` ends a pile, ``* *`` ends and discards
    it, ``- -`` ends the stream; lines that do not split into exactly two tokens are
    skipped.  The first sequence of a pile is the seed; it is stored as the target
    and -- unless its id shows up again -- once more as an ordinary read, exactly
    like the reference.  A pile is admitted when it holds at least ``min_n_read``
    sequences and its read bases cover the seed ``min_cov_aln`` times."""

    def __init__(self, stream, cfg: Settings, min_n_read: int, min_len_aln: int):
        self.stream, self.cfg = stream, cfg
        self.min_n_read, self.min_len_aln = min_n_read, min_len_aln

    def __iter__(self):
        pile, ids, bases, seed_id = [], set(), 0, None
        for line in self.stream:
            tok = line.split()
            if len(tok) != 2:
                continue
            name, seq = tok
            if len(seq) > MAX_SEQ_LEN:
                seq = seq[:MAX_SEQ_LEN - 1]
            if name in ("+", "*", "-"):
                if name == "-":
                    return
                if name == "+" and pile and len(pile) >= self.min_n_read and \
                        bases // len(pile[0]) >= self.cfg.min_cov_aln:
                    yield seed_id, select_reads(pile, self.cfg.max_n_read, self.cfg.max_cov_aln)
                pile, ids, bases, seed_id = [], set(), 0, None
                continue
            if len(seq) < self.min_len_aln:
                continue
            if not pile:
                pile.append(seq)
                seed_id = name
            if name not in ids:
                ids.add(name)
                pile.append(seq)
                bases += len(seq)


# --------------------------------------------------------------------------
# --trim (behaviour of consensus.py:48-99 and :123-146).  The k-mer window of every read
# comes from the GPU in one batch (GpuConsensus.trimmed -> k_trimwin.hip); the same
# window through the legacy table ABI (host structs, one read at a time) is kept for
# callers without an engine (tests with a stand-in consensus map).
# --------------------------------------------------------------------------
def window_from_range(rng, read_len, seed_len, edge_tolerance):
    """Post-processing of the raw aln_range (consensus.py:66-99): (q_start, q_end, score)
    or None."""
    q0, q1, t0, t1, score = rng
    pad = KMER + KMER // 2
    q1 = min(q1 + pad, read_len)
    t1 = min(t1 + pad, seed_len)
    if q0 > edge_tolerance and t0 > edge_tolerance:
        return None
    if read_len - q1 > edge_tolerance and seed_len - t1 > edge_tolerance:
        return None
    if q1 - q0 <= 500:
        return None
    return q0, q1, int(score * 48)


def mapped_window(kup, read, seed, edge_tolerance):
    """Chained k-mer span of ``read`` on ``seed``: (q_start, q_end, score) or None."""
    from ctypes import c_char_p
    rb, sb = read.encode("ascii"), seed.encode("ascii")
    table = kup.allocate_kmer_lookup(1 << (2 * KMER))
    codes = kup.allocate_seq(len(sb))
    chain = kup.allocate_seq_addr(len(sb))
    try:
        kup.add_sequence(0, KMER, c_char_p(sb), len(sb), chain, codes, table)
        kup.mask_k_mer(1 << (2 * KMER), table, 16)
        hits = kup.find_kmer_pos_for_seq(c_char_p(rb), len(rb), KMER, chain, table)
        rng = kup.find_best_aln_range2(hits, KMER, KMER * 50, 25)
        q0, q1, t0, t1, score = rng[0].s1, rng[0].e1, rng[0].s2, rng[0].e2, rng[0].score
        kup.free_kmer_match(hits)
        kup.free_aln_range(rng)
    finally:
        kup.free_seq_addr_array(chain)
        kup.free_seq_array(codes)
        kup.free_kmer_lookup(table)
    return window_from_range((q0, q1, t0, t1, score), len(read), len(seed), edge_tolerance)


def trimmed_pile(kup, pile, cfg: Settings, ranges=None):
    """``ranges``: the raw aln_range of every read of the pile (seed entry unused) when
    the GPU computed them, else they are computed one by one through ``kup``."""
    seed = pile[0]
    windows = []
    for j, read in enumerate(pile[1:], 1):
        if ranges is not None:
            w = window_from_range(ranges[j], len(read), len(seed), cfg.edge_tolerance)
        else:
            w = mapped_window(kup, read, seed, cfg.edge_tolerance)
        if w is None:
            continue
        q0, q1, score = w
        if score > 1000:
            q0 += cfg.trim_size
            q1 -= cfg.trim_size
            windows.append((q1 - q0, read[q0:q1]))
    windows.sort(key=lambda w: -w[0])  # longest window first, stable
    out = [seed] + [w[1] for w in windows]
    if len(out) - 1 > cfg.max_n_read:
        out = select_reads(out, cfg.max_n_read, cfg.max_cov_aln, presorted=True)
    return out


# --------------------------------------------------------------------------
# batching over one or more GPUs
# --------------------------------------------------------------------------
def ordered_parallel(fn, items, workers, window=None):
    """``fn`` over ``items`` on ``workers`` threads, results in input order (the shape of
    ``Pool.imap``, consensus.py:274); at most ``window`` items are out at any time."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    window = window or 2 * workers
    with ThreadPoolExecutor(max_workers=workers) as pool:
        out = collections.deque()
        for it in items:
            out.append(pool.submit(fn, it))
            if len(out) >= window:
                yield out.popleft().result()
        while out:
            yield out.popleft().result()


class GpuConsensus:
    """Ordered map over piles on the visible GPU(s).

    Piles are cut into batches of ``batch_bases``; every batch goes to the engine with the
    least work queued (falcon_amd/devices.py: independent work queues -- piles never
    interact, so there is no collective) and three batches per engine are in flight, the
    consensus stage of one beside the index, chaining and alignment of the next."""

    def __init__(self, min_cov, min_idt, engines=None, batch_bases=400_000_000, backend=None):
        from falcon_amd.devices import DevicePool, EngineBackend, SharedGpu, open_pool
        # (no engines given: ONE GPU, chosen so that the jobs fc_run starts together spread over
        # the node; another one only when this one's queue stays full and that one is idle)
        self.pool = open_pool() if engines is None else DevicePool(engines)
        self.shared = SharedGpu(self.pool, backend or EngineBackend(min_cov, min_idt))
        self.engines = self.shared.engines
        self.parallel = self.shared.parallel
        self.min_cov, self.min_idt = min_cov, min_idt
        self.batch_bases = batch_bases

    @staticmethod
    def _batches(piles, batch_bases):
        batch, bases = [], 0
        for p in piles:
            batch.append(p)
            bases += sum(map(len, p))
            if bases >= batch_bases:
                yield batch
                batch, bases = [], 0
        if batch:
            yield batch

    def imap(self, piles):
        """Consensus strings of an iterable of piles (lists of sequences), in order."""
        def one(batch):
            return self.shared.finish(self.shared.stage(batch))
        for res in ordered_parallel(one, self._batches(piles, self.batch_bases), self.parallel):
            yield from res

    # ---- --trim: windows of whole batches of piles on the GPU ----------------------
    def trimmed(self, piles, cfg):
        """Map piles -> trimmed piles (consensus.py:123-146), the k-mer windows of a batch
        of piles computed in one go on the least loaded engine."""
        def one(batch):
            dev = self.pool.take()
            try:
                b = dev.engine.batch(batch)
                try:
                    with dev.run_lock:
                        b.trim_windows(KMER, 16)
                    out, g = [], 0
                    for pile in batch:
                        rng = []
                        for _ in pile:
                            r = b.range(g)
                            rng.append((r["s1"], r["e1"], r["s2"], r["e2"], r["score"]))
                            g += 1
                        out.append(trimmed_pile(None, pile, cfg, rng))
                    return out
                finally:
                    b.free()
            finally:
                self.pool.give_back(dev)
        for res in ordered_parallel(one, self._batches(piles, self.batch_bases), len(self.engines)):
            yield from res

    # ---- native ingest: PileSets straight from falcon_amd/csrc/reader.cpp ----------
    def stage(self, ps):
        """Stage a PileSet on the least loaded engine (host memcpy into pinned memory, H2D,
        2-bit pack).  The PileSet may be recycled by its reader once this returns."""
        return self.shared.stage(ps)

    def finish(self, staged, fasta=None):
        """Run a staged batch and return its consensus strings in pile order (or, with
        ``fasta=(seed ids, mode)``, its FASTA text and failed piles: SharedGpu.finish)."""
        return self.shared.finish(staged, fasta)

    @property
    def native_fasta(self):
        return self.shared.native_fasta

    def close(self):
        self.pool.close()


# --------------------------------------------------------------------------
# output (behaviour of consensus.py:275-299)
# --------------------------------------------------------------------------
_SOLID = re.compile("[ACGT]+")


def fasta_records(seed_id, cns, output_full, output_multi):
    """The FASTA text the reference prints for one consensus string."""
    if len(cns) < 500:
        return ""
    if output_full:
        return ">%s_f\n%s\n" % (seed_id, cns)
    solid = _SOLID.findall(cns)
    if not solid:
        return ""
    if not output_multi:
        best = solid[0]
        for s in solid[1:]:  # the last of the longest, like sort()[-1]
            if len(s) >= len(best):
                best = s
        return ">%s\n%s\n" % (seed_id, best)
    text, k = [], 0
    for s in solid:
        if len(s) < 500:
            continue
        if k == 10:
            break
        text.append(">prolog/%s%01d/%d_%d\n" % (seed_id, k, 0, len(s)))
        text.extend(s[i:i + 80] + "\n" for i in range(0, len(s), 80))
        k += 1
    return "".join(text)


# Piles fail alone (falcon_amd.h: fa_batch_pile_error -- e.g. more usable reads than the GPU
# consensus stage handles, which takes --max-n-read above 1024): every other pile is
# corrected and printed, the failed ones are named on stderr and make the exit status 3
# (FALCON_AMD_SKIP_FAILED_PILES=1: status 0) -- the reference would have corrected them too.
FAILED_PILES = []


class FailedPiles(list):
    """A list of seed ids that also keeps why each pile failed (the server tells the job, whose stderr is not
    the one the worker's log goes to)."""

    def __init__(self):
        super().__init__()
        self.reasons = {}


def _note_failed(failed_piles, sid, reason):
    lst = FAILED_PILES if failed_piles is None else failed_piles
    lst.append(sid)
    if hasattr(lst, "reasons"):
        lst.reasons[sid] = str(reason)
    LOG.error("seed %s is not corrected: %s", sid, reason)


def note_failed_piles(ids, cns_all, failed_piles=None):
    """``failed_piles``: the list the seeds go to (one per stream in the multi-stream worker);
    default: the process-wide one ``main`` looks at."""
    for sid, cns in zip(ids, cns_all):
        reason = getattr(cns, "reason", None)
        if reason is not None:
            _note_failed(failed_piles, sid, reason)


def _stream_fd(stream):
    """File descriptor behind ``stream`` when it is an OS-level stream (the native reader
    takes it over and python must not have read from it), else None (StringIO & co)."""
    if os.environ.get("FALCON_AMD_PY_READER"):
        return None
    try:
        return getattr(stream, "buffer", stream).fileno()
    except (AttributeError, OSError, ValueError):
        return None


# Bases per batch on the native path.  Measured on 3072 E. coli-like piles of text
# (scripts/exp_batch_policy.py, profiles/r01_v8_batch_policy.txt): constant 0.4 G 3020 piles/s,
# constant 1.3 G 2076, doubling from 0.4 G to 1.3 G 1372-1679 -- larger batches run the
# kernels at a better rate, but the host buffers they need (text, pinned copy: page faults,
# pinning) cost more than that buys, and small batches pipeline sooner.
NATIVE_BATCH_BASES = 400_000_000
# Batches of text the ingest thread may be ahead of the staging thread (8 x ~0.4 GB at most; the
# devices take ~0.2 s to open, a file delivers ~4 GB in that time).
NATIVE_READ_AHEAD = 8


def batch_schedule(batch_bases, grow=None):
    """Bases of the stream's n-th batch.  Default: every batch ``batch_bases`` (0.4 G: on a block of 25 GB larger
    batches were slower, a block is over before they have paid for their buffers -- DESIGN.md 6a).
    ``grow`` = "FROM:FACTOR:MAX" (FALCON_AMD_BATCH_GROW): from batch FROM on every batch is FACTOR times the one
    before, up to MAX bases -- for streams that go on (a server's jobs, a long pipe): the GPU takes 3072 piles at
    52 k piles/s and 473 at 37 k.  The records leave in input order whatever the batches are."""
    if not grow:
        return lambda n: batch_bases
    start, factor, top = grow.split(":")
    start, factor, top = int(start), float(factor), int(float(top))
    if start < 0 or factor < 1.0 or top < batch_bases:
        raise ValueError("FALCON_AMD_BATCH_GROW=%r: FROM >= 0, FACTOR >= 1, MAX >= the batch size" % grow)

    def bases_of(n):
        if n < start:
            return batch_bases
        return int(min(float(top), batch_bases * factor ** min(n - start + 1, 64)))
    return bases_of


def _run_native(args, cfg, fd, gpu, stdout, batch_bases=None, failed_piles=None, leave_open=False):
    """The worker's pipeline, records leaving in input order (``failed_piles``: where the
    seeds of piles that failed alone are collected, default the process-wide list):

      ingest thread    native reader -> the next batch (``batch_bases`` bases) of the stream
      staging thread   that batch onto the device with the least work queued (``gpu.stage``),
                       while the ingest thread reads the one behind it
      runner threads   ``gpu.parallel`` of them (three per engine): a batch's GPU stages
                       (``gpu.finish``: submit -- index, chaining and alignment queued under the
                       engine's lock -- then wait, while the consensus stage runs beside the
                       next batch's kernels) and the download
      printer thread   batches back in input order -> FASTA text

    ``gpu``: ``stage(pileset) -> handle`` (``handle.free()`` drops it), ``finish(handle) ->
    consensus strings``, optionally ``parallel`` -- or a function without arguments that
    returns such an object: it is called once the ingest thread is reading, so that opening
    the devices (HIP start-up) and the first two batches of the stream overlap."""
    import queue
    from falcon_amd.engine import Reader
    reader = Reader(fd, args.min_n_read, args.min_len_aln, cfg.min_cov_aln, cfg.max_n_read,
                    cfg.max_cov_aln)
    if batch_bases is None:
        batch_bases = int(os.environ.get("FALCON_AMD_BATCH_BASES", NATIVE_BATCH_BASES))
    bases_of = batch_schedule(batch_bases, os.environ.get("FALCON_AMD_BATCH_GROW"))
    failed = []
    stop = threading.Event()
    END = object()

    def fail(exc):
        failed.append(exc)
        stop.set()

    # What the reader hands out lives through the next `ahead` - 1 calls of reader.next() (that
    # many text buffers take turns): `lease` counts the buffers the stager has not finished
    # with.  The ingest thread runs ahead while the devices are still being opened and whenever
    # the stream comes faster than the GPU takes it (a block's text in a file); a pipe from
    # LA4Falcon never gets ahead, and buffers never filled are never touched.
    ahead = max(2, min(64, int(os.environ.get("FALCON_AMD_READ_AHEAD", NATIVE_READ_AHEAD))))
    reader.keep(ahead)
    raw = queue.Queue(maxsize=ahead - 1)
    lease = threading.Semaphore(ahead)

    def ingest():
        seq = 0
        try:
            while not stop.is_set():
                while not lease.acquire(timeout=0.1):
                    if stop.is_set():
                        return
                t0 = time.perf_counter()
                ps = reader.next(0, bases_of(seq))
                if ps is None:
                    break
                LOG.debug("t=%.3f ingest: batch %d, %d piles read in %.3f s", _clock(), seq, ps.n_pile,
                          time.perf_counter() - t0)
                raw.put((seq, ps))
                seq += 1
        except Exception as exc:
            fail(exc)
        finally:
            raw.put(END)

    stagers_left = [0]  # (set when the threads are made; the last stager to leave ends the runners' queue)
    stagers_lock = threading.Lock()
    staged_seqs, next_back = set(), [0]

    def give_back(seq):
        # The reader's promise counts CALLS (a batch lives through the next `ahead` - 1 of them), so its
        # buffers go back in stream order: a batch staged before an earlier one waits for that one.
        with stagers_lock:
            staged_seqs.add(seq)
            while next_back[0] in staged_seqs:
                staged_seqs.remove(next_back[0])
                next_back[0] += 1
                lease.release()

    def stager():
        try:
            while True:
                item = raw.get()
                if item is END:
                    raw.put(END)  # the other stagers see it too
                    return
                seq, ps = item
                try:
                    if stop.is_set():
                        continue
                    t0 = time.perf_counter()
                    out = (seq, (ps.seed_ids, ps.seed_ids_raw), gpu.stage(ps))
                    LOG.debug("t=%.3f stager: batch %d staged in %.3f s", _clock(), seq,
                              time.perf_counter() - t0)
                finally:
                    give_back(seq)
                staged.put(out)
        except Exception as exc:
            fail(exc)
            while True:  # (the reader never blocks on a full queue)
                item = raw.get()
                if item is END:
                    raw.put(END)
                    break
                give_back(item[0])
        finally:
            with stagers_lock:
                stagers_left[0] -= 1
                last = stagers_left[0] == 0
            if last:
                staged.put(END)

    def runner():
        try:
            while True:
                item = staged.get()
                if item is END:
                    staged.put(END)  # the other runners see it too
                    return
                seq, (ids, ids_raw), handle = item
                if stop.is_set():
                    getattr(handle, "free", lambda: None)()
                    continue
                t0 = time.perf_counter()
                if native_fasta:  # (text, [(pile, reason)]): no per-pile python object
                    res = gpu.finish(handle, (ids_raw, fasta_mode))
                else:
                    res = gpu.finish(handle)
                LOG.debug("t=%.3f runner: batch %d, GPU stages + download %.3f s", _clock(), seq,
                          time.perf_counter() - t0)
                done.put((seq, ids, res))
        except Exception as exc:
            fail(exc)
            # keep draining so that the ingest thread never blocks on a full queue
            while True:
                item = staged.get()
                if item is END:
                    staged.put(END)
                    return
                getattr(item[2], "free", lambda: None)()

    printed = []  # (time, piles) of every batch as it leaves, in order

    def printer():
        waiting, want = {}, 0
        try:
            while True:
                item = done.get()
                if item is END:
                    return
                waiting[item[0]] = item
                while want in waiting:
                    t0 = time.perf_counter()
                    _, ids, res = waiting.pop(want)
                    want += 1
                    if stop.is_set():
                        continue
                    if native_fasta:
                        text, bad = res
                        for p, reason in bad:
                            _note_failed(failed_piles, ids[p], reason)
                        write_bytes(text)
                        printed.append((time.perf_counter(), len(ids)))
                        LOG.debug("t=%.3f printer: %d piles in %.3f s", _clock(), len(ids),
                                  time.perf_counter() - t0)
                        continue
                    cns_all = res
                    note_failed_piles(ids, cns_all, failed_piles)
                    stdout.write("".join(fasta_records(sid, cns, args.output_full, args.output_multi)
                                         for sid, cns in zip(ids, cns_all)))
                    printed.append((time.perf_counter(), len(ids)))
                    LOG.debug("t=%.3f printer: %d piles in %.3f s", _clock(), len(ids),
                              time.perf_counter() - t0)
        except Exception as exc:  # (a closed stdout, say): stop the pipeline, report below
            fail(exc)
            while done.get() is not END:
                pass

    t_in = threading.Thread(target=ingest, daemon=True)
    t_in.start()
    if callable(gpu):
        try:
            gpu = gpu()
        except BaseException:
            stop.set()
            while t_in.is_alive() or not raw.empty():  # (the reader's batches are its own)
                try:
                    raw.get(timeout=0.05)
                except queue.Empty:
                    pass
            reader.close()
            raise
    n_run = max(1, int(getattr(gpu, "parallel", 1)))
    # the output rules in native code (fa_batch_fasta) when the engine offers them
    native_fasta = bool(getattr(gpu, "native_fasta", False)) and not os.environ.get("FALCON_AMD_PY_PRINTER")
    fasta_mode = 2 if args.output_full else 1 if args.output_multi else 0
    raw_out = getattr(stdout, "buffer", None)

    def write_bytes(text):
        if raw_out is not None:
            stdout.flush()
            raw_out.write(text)
        else:
            stdout.write(text.decode("utf-8", "replace"))

    staged = queue.Queue(maxsize=n_run)
    done = queue.Queue(maxsize=2 * n_run + 2)
    # (FALCON_AMD_STAGERS=2: a batch's layout and device buffers beside the packing and upload of the one before
    # -- those two hold the context's one pinned staging buffer; the printer puts the batches back in order.
    # Measured: nothing, profiles/r05_e2e_stagers.txt -- the stream waits for the GPU's batches, not for staging)
    n_stage = max(1, min(4, int(os.environ.get("FALCON_AMD_STAGERS", "1"))))
    stagers_left[0] = n_stage
    t_stage = [threading.Thread(target=stager, daemon=True) for _ in range(n_stage)]
    t_run = [threading.Thread(target=runner, daemon=True) for _ in range(n_run)]
    t_out = threading.Thread(target=printer, daemon=True)
    for t in t_stage + [t_out] + t_run:
        t.start()
    try:
        for t in t_run:
            t.join()
    except BaseException:
        stop.set()
        raise
    finally:
        LOG.debug("t=%.3f runners done", _clock())
        done.put(END)   # (behind everything the runners delivered)
        t_out.join()
        stop.set()
        LOG.debug("t=%.3f printer done", _clock())
        # the reader may only be closed once the ingest and staging threads have left it;
        # what they had staged meanwhile is released
        while t_in.is_alive() or any(t.is_alive() for t in t_stage):
            try:
                item = staged.get(timeout=0.05)
            except queue.Empty:
                continue
            if item is not END:
                getattr(item[2], "free", lambda: None)()
        if leave_open and not failed:  # (the process is about to end, _leave: nothing is unmapped or freed first --
            _LEFT_OPEN.append(reader)   # not by a destructor either)
        else:
            reader.close()
    LOG.debug("t=%.3f stream finished", _clock())
    if len(printed) >= 3:
        # the worker's own steady state: from the first batch leaving to the last one (what
        # start-up -- process, HIP, first buffers -- and exit add is the rest of the wall time)
        span = printed[-1][0] - printed[0][0]
        n_after = sum(n for _, n in printed[1:])
        if span > 0:
            LOG.info("falcon_amd consensus: %d piles in %d batches; steady state %.0f piles/s (%d piles in %.3f s "
                     "between the first and the last batch printed)", sum(n for _, n in printed), len(printed),
                     n_after / span, n_after, span)
    if failed:
        raise failed[0]


def run(args, stdin=None, stdout=None, consensus_map=None, leave_open=False):
    """``consensus_map`` (tests) replaces the GPU map: iterable of piles -> iterable of
    consensus strings."""
    stdin = sys.stdin if stdin is None else stdin
    stdout = sys.stdout if stdout is None else stdout
    logging.basicConfig(level=int(round(10 * args.verbose_level)))
    if args.max_n_read > 1024:
        # (beyond 1024 alignments over one stretch of a seed a pile is
        # corrected on a best-effort basis and fails ALONE, exit status 3, when it cannot be)
        LOG.warning("--max-n-read %d: piles are guaranteed to be corrected up to 1024 overlapping reads; "
                    "deeper ones that cannot be are reported and skipped", args.max_n_read)
    # the reference refuses more workers than cores (consensus.py:258); callers
    # (consensus_task.py:44-54) clamp --n-core accordingly, so keep the contract
    assert args.n_core <= multiprocessing.cpu_count(), \
        'Requested n_core={} > cpu_count={}'.format(args.n_core, multiprocessing.cpu_count())
    cfg = settings_from(args)
    kup = None
    gpu = None
    opened = []

    def open_gpu():
        opened.append(GpuConsensus(args.min_cov, args.min_idt))
        from falcon_amd.devices import held_slots
        LOG.info("falcon_amd consensus on %d engine(s), device(s) %s, lock slot(s) %s (t=%.3f)", len(opened[0].engines),
                 ",".join(str(getattr(e, "device", "?")) for e in opened[0].engines),
                 ",".join("%d.%d" % ds for ds in held_slots) or "none", _clock())
        return opened[0]

    fd = _stream_fd(stdin)
    if consensus_map is None and not args.trim and fd is not None:
        ok = False
        try:
            _run_native(args, cfg, fd, open_gpu, stdout, leave_open=leave_open)
            ok = True
        finally:
            stdout.flush()
            if leave_open and ok:
                _LEFT_OPEN.extend(opened)
            else:
                for g in opened:
                    g.close()
                LOG.debug("t=%.3f engines closed", _clock())
        return
    if consensus_map is None:
        gpu = open_gpu()
        consensus_map = gpu.imap

    if args.trim and gpu is None:
        from falcon_amd import falcon_kit as fk
        kup = fk.kup
    seed_ids = []

    def piles():
        for seed_id, pile in PileReader(stdin, cfg, args.min_n_read, args.min_len_aln):
            seed_ids.append(seed_id)
            yield trimmed_pile(kup, pile, cfg) if kup is not None else pile

    todo = piles()
    if args.trim and gpu is not None:
        todo = gpu.trimmed(todo, cfg)
    try:
        for i, cns in enumerate(consensus_map(todo)):
            note_failed_piles(seed_ids[i:i + 1], [cns])
            stdout.write(fasta_records(seed_ids[i], cns, args.output_full, args.output_multi))
    finally:
        if gpu is not None:
            gpu.close()
    stdout.flush()


def main(argv=None, own_process=False):
    """The reference's entry point (falcon_kit/mains/consensus.py:302, setup.py:51).  own_process: the
    caller IS the process (``console_main``) -- reader and engines are then left as they are and the
    process ends without tearing anything down."""
    args = parse_args(sys.argv if argv is None else argv)
    LOG.debug("t=%.3f arguments parsed", _clock())
    fast = own_process and not os.environ.get("FALCON_AMD_SLOW_EXIT")
    run(args, leave_open=fast)
    LOG.debug("t=%.3f run() returned", _clock())
    code = 0
    if FAILED_PILES and not os.environ.get("FALCON_AMD_SKIP_FAILED_PILES"):
        sys.stderr.write("falcon_amd: %d pile(s) were not corrected (see above)\n" % len(FAILED_PILES))
        code = 3
    if fast:
        _leave(code)
    if code:
        sys.exit(code)


def console_main(argv=None):
    """``python -m falcon_amd.mains.consensus``, ``bin/fc_consensus``, the drop-in's
    ``python -m falcon_kit.mains.consensus``: main() in a process of its own."""
    if os.environ.get("FALCON_AMD_SERVER"):
        # a long-lived worker on this node (falcon_amd/mains/consensus_server.py) takes the stream: this
        # process hands over its descriptors and waits -- or goes on by itself when nobody is there
        from falcon_amd.mains._serve_client import try_server
        code = try_server(sys.argv if argv is None else argv)
        if code is not None:
            sys.stdout.flush()
            os._exit(code)
    main(argv, own_process=True)


def _leave(code):
    """The console command's exit: everything is written and flushed, and the process is its own --
    so it ends here, without the interpreter's and the HIP runtime's teardown (threads, module
    finalisers, unmapping gigabytes of text buffers the kernel frees anyway: ~0.15 s of a block's
    1.6 s).  FALCON_AMD_SLOW_EXIT=1: the ordinary way out."""
    try:
        sys.stdout.flush()
        sys.stderr.flush()
        logging.shutdown()
    finally:
        os._exit(code)


if __name__ == "__main__":
    console_main(sys.argv)
