"""A consensus worker that stays: one process per node holds the GPUs, and every ``fc_consensus`` /
``python -m falcon_amd.mains.consensus`` started on the node hands it its stdin and stdout.

The reference starts one consensus process per ``.las`` block (falcon_kit/mains/consensus_task.py:90, one job
record per block: consensus_split.py:55-85), and so does a drop-in for it -- but here a process start is an
interpreter, a HIP context, 7 GB of arena, pinned staging buffers and the code objects: 0.45 s of the 1.6 s a
25 GB block takes, and the VRAM a process gives back is wiped by the driver before the next one gets it.  With
a server on the node that is paid once:

    python -m falcon_amd.mains.consensus_server --socket $TMPDIR/falcon_amd.sock &      # once per node
    export FALCON_AMD_SERVER=$TMPDIR/falcon_amd.sock                                    # in the jobs' environment
    LA4Falcon -H$CUTOFF -fo db las | python -m falcon_amd.mains.consensus <opts> > cns.fasta   # unchanged

The job's process connects to the socket, sends its command line and -- over SCM_RIGHTS -- its descriptors 0, 1
and 2, waits, and exits with the status the server reports (falcon_amd/mains/_serve_client.py: standard library
only, ~30 ms).  The server runs the worker's own pipeline on the descriptors (consensus._run_native: native
reader, staging, runner and printer threads) with the engines it holds; jobs that come at the same time share
the devices' work queues like the jobs of consensus_multi do (a batch goes to the device with the least work
queued; piles of different jobs never interact; every job's records keep their input order).  The bytes a job
gets are the bytes the stand-alone worker prints; piles that fail alone are named on the job's stderr and make
its status 3 (FALCON_AMD_SKIP_FAILED_PILES as ever).  A job the server cannot take (``--trim``; the server
gone; no socket) runs in its own process as before: FALCON_AMD_SERVER never makes a job fail.

There is no CPU fallback: without libfalcon_amd.so and a HIP device the server does not start.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import signal
import socket
import sys
import threading
import time

from falcon_amd.mains import consensus as single

LOG = logging.getLogger("falcon_amd.consensus_server")


def _handle(conn, pool, stats, make_backend=None):
    from falcon_amd.devices import EngineBackend, SharedGpu
    make_backend = make_backend or (lambda a: EngineBackend(a.min_cov, a.min_idt))
    fds = []
    reply = {"status": 1, "message": "internal error"}
    try:
        msg, fds, _flags, _addr = socket.recv_fds(conn, 1 << 16, 3)
        req = json.loads(msg.decode())
        if len(fds) != 3:
            raise ValueError("three descriptors expected, %d received" % len(fds))
        args = single.parse_args(req["argv"])
        if args.trim:
            reply = {"status": "decline", "message": "--trim runs in the job's own process"}
            return
        cfg = single.settings_from(args)
        failed = single.FailedPiles()
        t0 = time.perf_counter()
        out = os.fdopen(os.dup(fds[1]), "w")
        try:
            single._run_native(args, cfg, fds[0], SharedGpu(pool, make_backend(args)), out, failed_piles=failed)
        finally:
            out.flush()
            out.close()
        code = 0
        if failed:
            with os.fdopen(os.dup(fds[2]), "w") as err:
                for sid in failed:
                    err.write("falcon_amd: seed %s is not corrected: %s\n" % (sid, failed.reasons.get(sid, "?")))
                err.write("falcon_amd: %d pile(s) were not corrected (see above)\n" % len(failed))
            if not req.get("skip_failed"):
                code = 3
        reply = {"status": code, "seconds": round(time.perf_counter() - t0, 3)}
        with stats["lock"]:
            stats["jobs"] += 1
    except BaseException as exc:  # a job that fails does not take the server with it
        LOG.error("job failed: %r", exc)
        reply = {"status": 1, "message": repr(exc)}
    finally:
        for fd in fds:
            try:
                os.close(fd)
            except OSError:
                pass
        try:
            conn.sendall(json.dumps(reply).encode())
        except OSError:
            pass
        conn.close()
        with stats["lock"]:
            stats["active"] -= 1
            stats["last"] = time.monotonic()


def serve(path, idle_exit=0.0, ready=None, pool=None, make_backend=None, stop=None):
    """Listen on the Unix socket `path` until SIGTERM / SIGINT, or until no job has come for `idle_exit`
    seconds (0: for good).  `ready`: a file object that gets one line once jobs can connect.
    (`pool`, `make_backend`, `stop`: the CPU tests' stand-ins for the devices, and their way to end it.)"""
    from falcon_amd.devices import DevicePool, open_engines
    if pool is None:
        pool = DevicePool(open_engines(all_devices=os.environ.get("FALCON_AMD_DEVICES") is None))
    # a socket file somebody still answers on is a live server: its jobs must not lose it to a second one
    probe = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        probe.settimeout(1.0)
        probe.connect(path)
        alive = True
    except OSError:
        alive = False
    finally:
        probe.close()
    if alive:
        if pool is not None and hasattr(pool, "close"):
            pool.close()
        raise RuntimeError("falcon_amd: a consensus server is already listening on %s" % path)
    try:
        os.unlink(path)
    except OSError:
        pass
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    old = os.umask(0o077)   # (whoever can connect gets this user's GPUs and writes through this user's process)
    try:
        srv.bind(path)
    finally:
        os.umask(old)
    srv.listen(64)
    srv.settimeout(0.5)
    stats = {"lock": threading.Lock(), "active": 0, "jobs": 0, "last": time.monotonic()}
    stop = stop or threading.Event()
    for s in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(s, lambda *_: stop.set())
        except ValueError:  # (not the main thread: tests)
            pass
    LOG.info("falcon_amd consensus server on %d GPU(s), socket %s", len(pool.devices), path)
    if ready is not None:
        ready.write("falcon_amd consensus server ready on %s (%d GPU(s))\n" % (path, len(pool.devices)))
        ready.flush()
    try:
        while not stop.is_set():
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                with stats["lock"]:
                    idle = stats["active"] == 0 and time.monotonic() - stats["last"] > idle_exit > 0
                if idle:
                    LOG.info("no job for %.0f s: leaving", idle_exit)
                    break
                continue
            conn.settimeout(None)
            with stats["lock"]:
                stats["active"] += 1
            threading.Thread(target=_handle, args=(conn, pool, stats, make_backend), daemon=True).start()
        # jobs that are running finish
        while True:
            with stats["lock"]:
                if stats["active"] == 0:
                    break
            time.sleep(0.05)
    finally:
        srv.close()
        try:
            os.unlink(path)
        except OSError:
            pass
        pool.close()
    return stats["jobs"]


def main(argv=None):
    ap = argparse.ArgumentParser(description="long-lived falcon_sense worker for the jobs of one node")
    ap.add_argument("--socket", default=os.environ.get("FALCON_AMD_SERVER") or
                    os.path.join(os.environ.get("TMPDIR", "/tmp"), "falcon_amd.%d.sock" % os.getuid()))
    ap.add_argument("--idle-exit", type=float, default=0.0, help="leave after this many seconds without a job (0: never)")
    ap.add_argument("-v", "--verbose-level", type=float, default=2.0)
    args = ap.parse_args((sys.argv if argv is None else argv)[1:])
    logging.basicConfig(level=int(round(10 * args.verbose_level)))
    serve(args.socket, args.idle_exit, ready=sys.stdout)


if __name__ == "__main__":
    main()
