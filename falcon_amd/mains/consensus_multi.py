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


# the devices, their work queues and what a job does with one: falcon_amd/devices.py
from falcon_amd.devices import (Device, DevicePool, EngineBackend, SharedGpu,  # noqa: E402,F401
                                Staged as _Staged, open_engines)


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
    not_corrected = []  # this job's piles that failed alone (named on stderr by the worker)
    try:
        try:
            with open(tmp, "w") as out:
                single._run_native(args, cfg, fd, SharedGpu(pool, backend), out,
                                   failed_piles=not_corrected)
        finally:
            close()  # (a producer that failed fails the job)
        # A FASTA that lacks reads the reference would have corrected is not a complete
        # output: the job fails like the single-stream worker exits 3 (same opt-out).
        if not_corrected and not os.environ.get("FALCON_AMD_SKIP_FAILED_PILES"):
            raise RuntimeError("%d pile(s) were not corrected: %s" % (
                len(not_corrected), ", ".join(not_corrected[:8]) + (" ..." if len(not_corrected) > 8 else "")))
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
        # engines (contexts, each with its own streams and work slots) per GPU:
        # FALCON_AMD_ENGINES_PER_DEVICE (1: an engine already keeps two batches in flight, the
        # sequential stages of one beside the throughput stages of the next)
        pool = DevicePool(open_engines(all_devices=True))  # (several streams: they can feed every GPU)
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
