"""bench.py, the end-to-end legs: LA4Falcon text in -> FASTA out through the consensus workers in processes of their own
(SURVEY.md 8d "end-to-end"); reported beside `value`, never as it."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import tempfile
import time

from benchlib.workloads import write_la4falcon

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


E2E_REPEATS = 10  # the end-to-end stream = the step's piles this many times: 30 720 piles, so


def end_to_end(piles, extra_args=(), expect=None, repeats=E2E_REPEATS):
    """SURVEY.md 8d "end-to-end": LA4Falcon text on stdin -> FASTA on stdout through the
    consensus worker (falcon_amd.mains.consensus: native reader, staging, GPU stages,
    printing) in a process of its own, on the piles of this workload written out as text.
    Reported beside `value` (kernel-only, inputs resident in HBM), never as it.
    `expect`: consensus strings of these piles from the resident batch -- the FASTA must be
    what the output rules (consensus.py:275-299) make of them, byte for byte."""
    root = ROOT
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "piles.txt"), os.path.join(tmp, "cns.fasta")
        # (as many repeats as the scratch directory holds with room to spare: ~0.83 MB of text per pile)
        import shutil
        per_repeat = sum(sum(len(x) + 10 for x in p) for p in piles) + 1
        repeats = max(1, min(repeats, int(shutil.disk_usage(tmp).free * 0.6 // per_repeat)))
        with open(src, "wb") as f:
            write_la4falcon(piles, f, repeats)
        size = os.path.getsize(src)
        cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt",
               "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"] + list(extra_args) + \
            os.environ.get("FALCON_BENCH_E2E_ARGS", "").split()
        # (a profiler wrapped around this process stays with this process: the worker's
        # launches are at another batch size and would blur its per-kernel averages)
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))}
        # three workers back to back, the median counts: fc_run starts one consensus process
        # per .las block one after the other, so a worker that starts right behind one that
        # released its VRAM (amdgpu wipes it) IS production (all three are listed)
        walls, steady = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            with open(src) as fin, open(dst, "w") as fout:
                p = subprocess.run(cmd, stdin=fin, stdout=fout, stderr=subprocess.PIPE, check=True, cwd=root,
                                   timeout=600, env=env, text=True)
            walls.append(time.perf_counter() - t0)
            for ln in p.stderr.split("\n"):  # the worker's own report (consensus._run_native)
                if "steady state" in ln:
                    steady.append(float(ln.split("steady state")[1].split()[0]))
        wall = sorted(walls)[1]  # the MEDIAN of three back-to-back workers (all listed)
        with open(dst) as f:
            text = f.read()
        bases = sum(len(ln) for ln in text.split("\n") if not ln.startswith(">"))
    n = repeats * len(piles)
    out = {"piles_per_sec": round(n / wall, 1), "text_MB_per_sec": round(size / 1e6 / wall, 1),
           "fasta_bases_per_sec": round(bases / wall, 1), "wall_s": round(wall, 2),
           "runs_wall_s": [round(w, 2) for w in walls],
           "worker_steady_state_piles_per_sec": steady,
           "what": "%d piles (the step's %d, %d times; %.0f MB of text from the page cache) -> FASTA, one "
                   "worker process on one GPU, process start and HIP initialisation included; three workers "
                   "back to back, the median wall time counts (`worker_steady_state_piles_per_sec`: what each "
                   "worker reports between its first and its last batch printed)"
                   % (n, len(piles), repeats, size / 1e6)}
    if expect is not None:
        from falcon_amd.mains.consensus import fasta_records
        want = "".join(fasta_records("%09d" % (rep * len(piles) + i), c, False, True)
                       for rep in range(repeats) for i, c in enumerate(expect))
        out["fasta_identical_to_resident_batch"] = (want == text)
        out["fasta_sha1"] = hashlib.sha1(text.encode()).hexdigest()[:16]
    return out


def end_to_end_multi(piles, n_streams, repeats=E2E_REPEATS):
    """N > 1: the same text as `n_streams` jobs of ONE multi-stream worker process
    (falcon_amd.mains.consensus_multi) over all visible GPUs -- how a node is fed: a single
    stream has one reader and one staging thread, which one device's batches already keep
    busy.  Every job's FASTA must equal what the single-stream worker prints
    for the same text on one GPU (run afterwards, outside the timing)."""
    root = ROOT
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "piles.txt")
        # (as many repeats as the scratch directory holds with room to spare, like end_to_end)
        import shutil
        per_repeat = sum(sum(len(x) + 10 for x in p) for p in piles) + 1
        repeats = max(1, min(repeats, int(shutil.disk_usage(tmp).free * 0.6 // per_repeat)))
        with open(src, "wb") as f:
            write_la4falcon(piles, f, repeats)
        size = os.path.getsize(src)
        opts = ["--output-multi", "--min-idt", "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))
               and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.pop("FALCON_AMD_DEVICES", None)
        jobs = []
        for j in range(n_streams):
            jobs += ["--job", src, os.path.join(tmp, "cns_%d.fasta" % j)]
        cmd = [sys.executable, "-m", "falcon_amd.mains.consensus_multi"] + opts + jobs
        walls = []
        for _ in range(2):
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, cwd=root, timeout=max(900, int(60 + n_streams * size / 2e8)), env=env)
            walls.append(time.perf_counter() - t0)
        ref = os.path.join(tmp, "single.fasta")
        with open(src) as fin, open(ref, "w") as fout:
            subprocess.run([sys.executable, "-m", "falcon_amd.mains.consensus"] + opts, stdin=fin, stdout=fout,
                           check=True, cwd=root, timeout=900, env=dict(env, FALCON_AMD_DEVICES="0"))
        want = open(ref).read()
        same = all(open(os.path.join(tmp, "cns_%d.fasta" % j)).read() == want for j in range(n_streams))
    n = repeats * len(piles) * n_streams
    walls.sort()
    wall = walls[len(walls) // 2]
    return {"piles_per_sec": round(n / wall, 1), "text_MB_per_sec": round(size * n_streams / 1e6 / wall, 1),
            "wall_s": round(wall, 2), "runs_wall_s": [round(w, 2) for w in walls], "streams": n_streams,
            "every_stream_identical_to_the_single_stream_worker": bool(same),
            "what": "%d streams of %d piles each (%.0f MB of text each, from the page cache) -> %d FASTA files, "
                    "one multi-stream worker process over all visible GPUs, process start included; the slower "
                    "of two runs counts" % (n_streams, repeats * len(piles), size / 1e6, n_streams)}


def end_to_end_workers(piles, n_workers, repeats=E2E_REPEATS, worker_cmd=None):
    """N > 1, the way fc_run feeds a node: `n_workers` single-stream consensus jobs started at the same
    moment (consensus_split.py:55-85 runs one per LA4Falcon block, several at a time), every one a
    process of its own that finds a GPU for itself through the lock slots of falcon_amd/devices.py.
    Per worker: the device it took, its wall time (process start to exit), piles/s and text rate; the
    aggregate counts the slowest.  Every FASTA must be the same text."""
    import re
    root = ROOT
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "piles.txt")
        import shutil
        per_repeat = sum(sum(len(x) + 10 for x in p) for p in piles) + 1
        repeats = max(1, min(repeats, int(shutil.disk_usage(tmp).free * 0.6 // per_repeat)))
        with open(src, "wb") as f:
            write_la4falcon(piles, f, repeats)
        size = os.path.getsize(src)
        # (worker_cmd: a stand-in for the worker, tests/test_bench_ranks.py -- the plumbing runs without a GPU)
        cmd = worker_cmd or [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70",
                             "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))
               and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                             "FALCON_AMD_DEVICES", "FALCON_AMD_DEVICE")}
        env["FALCON_AMD_LOCK_DIR"] = os.path.join(tmp, "slots")  # (nobody else's jobs in this count)
        os.makedirs(env["FALCON_AMD_LOCK_DIR"])
        t0 = time.perf_counter()
        procs = []
        for j in range(n_workers):
            fin, fout = open(src), open(os.path.join(tmp, "cns_%d.fasta" % j), "w")
            ferr = open(os.path.join(tmp, "err_%d.txt" % j), "w+")
            procs.append((subprocess.Popen(cmd, stdin=fin, stdout=fout, stderr=ferr, cwd=root, env=env), fin,
                          fout, ferr))
        wall = [None] * n_workers
        deadline = t0 + max(900, 60 + n_workers * size / 2e8)
        while any(w is None for w in wall):
            for j, (p, *_f) in enumerate(procs):
                if wall[j] is None and p.poll() is not None:
                    wall[j] = time.perf_counter() - t0
            if time.perf_counter() > deadline:
                for p, *_f in procs:
                    if p.poll() is None:
                        p.kill()
                raise RuntimeError("end_to_end_workers: a worker did not finish")
            time.sleep(0.005)
        workers, texts = [], []
        for j, (p, fin, fout, ferr) in enumerate(procs):
            fin.close()
            fout.close()
            ferr.seek(0)
            log = ferr.read()
            ferr.close()
            if p.returncode != 0:
                raise RuntimeError("end_to_end_workers: worker %d exited %d: %s" % (j, p.returncode, log[-400:]))
            dev = re.search(r"device\(s\) ([0-9]+(?:,[0-9]+)*)", log)
            slot = re.search(r"lock slot\(s\) ([0-9.,]+)", log)
            steady = re.search(r"steady state ([0-9.]+) piles/s", log)
            texts.append(open(os.path.join(tmp, "cns_%d.fasta" % j)).read())
            workers.append({"worker": j, "devices": dev.group(1) if dev else None,
                            "lock_slots": slot.group(1) if slot else None, "wall_s": round(wall[j], 2),
                            "piles_per_sec": round(repeats * len(piles) / wall[j], 1),
                            "text_GB_per_sec": round(size / 1e9 / wall[j], 2),
                            "steady_state_piles_per_sec": float(steady.group(1)) if steady else None})
    n = repeats * len(piles) * n_workers
    slowest = max(wall)
    return {"piles_per_sec": round(n / slowest, 1), "text_MB_per_sec": round(size * n_workers / 1e6 / slowest, 1),
            "wall_s": round(slowest, 2), "workers": workers,
            "distinct_devices": len({w["devices"] for w in workers}),
            "distinct_lock_slots": len({w["lock_slots"] for w in workers if w["lock_slots"]}),
            "every_fasta_identical": all(t == texts[0] for t in texts) and len(texts[0]) > 0,
            "what": "%d single-stream workers started together, %d piles each (%.0f MB of text each, from the page "
                    "cache) -> %d FASTA files; every worker is a process of its own that takes a GPU through the "
                    "lock slots (falcon_amd/devices.py), process start and HIP initialisation included; the "
                    "aggregate counts the slowest worker" % (n_workers, repeats * len(piles), size / 1e6, n_workers)}


def end_to_end_served(piles, expect=None, repeats=E2E_REPEATS, jobs=3):
    """The same stream as end_to_end(), but through a worker that STAYS (falcon_amd.mains.consensus_server): the
    server is started first and is not on the clock -- a node starts it once -- and then `jobs` job processes
    (`python -m falcon_amd.mains.consensus` with FALCON_AMD_SERVER set: interpreter start, connect, hand over stdin /
    stdout, wait) run back to back, each on the whole stream.  What a .las block costs when process start, HIP
    context, arena and the driver's VRAM wipe are paid once per node instead of once per block."""
    import shutil
    import signal
    with tempfile.TemporaryDirectory() as tmp:
        src, dst, sock = os.path.join(tmp, "piles.txt"), os.path.join(tmp, "cns.fasta"), os.path.join(tmp, "srv.sock")
        per_repeat = sum(sum(len(x) + 10 for x in p) for p in piles) + 1
        repeats = max(1, min(repeats, int(shutil.disk_usage(tmp).free * 0.6 // per_repeat)))
        with open(src, "wb") as f:
            write_la4falcon(piles, f, repeats)
        size = os.path.getsize(src)
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))}
        t_up = time.perf_counter()
        srv = subprocess.Popen([sys.executable, "-m", "falcon_amd.mains.consensus_server", "--socket", sock], cwd=ROOT,
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            line = srv.stdout.readline()
            if "ready" not in line:
                raise RuntimeError("the consensus server did not come up: %s %s" % (line, srv.stderr.read()[-400:]))
            t_up = time.perf_counter() - t_up
            cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70", "--min-cov",
                   "4", "--max-n-read", "200", "--n-core", "1"]
            walls = []
            for _ in range(jobs):
                t0 = time.perf_counter()
                with open(src) as fin, open(dst, "w") as fout:
                    subprocess.run(cmd, stdin=fin, stdout=fout, check=True, cwd=ROOT, timeout=600,
                                   env=dict(env, FALCON_AMD_SERVER=sock))
                walls.append(time.perf_counter() - t0)
            with open(dst) as f:
                text = f.read()
        finally:
            srv.send_signal(signal.SIGTERM)
            try:
                srv.wait(timeout=60)
            except subprocess.TimeoutExpired:
                srv.kill()
    n = repeats * len(piles)
    wall = sorted(walls)[len(walls) // 2]
    out = {"piles_per_sec": round(n / wall, 1), "text_MB_per_sec": round(size / 1e6 / wall, 1), "wall_s": round(wall, 2),
           "runs_wall_s": [round(w, 2) for w in walls], "server_start_s": round(t_up, 2),
           "what": "%d piles (%.0f MB of text) -> FASTA per job, %d job processes back to back handing their stdin / "
                   "stdout to one consensus server on the node (started before, %.2f s, not on the clock); the median "
                   "job counts, its own process start included" % (n, size / 1e6, jobs, t_up)}
    if expect is not None:
        from falcon_amd.mains.consensus import fasta_records
        want = "".join(fasta_records("%09d" % (rep * len(piles) + i), c, False, True)
                       for rep in range(repeats) for i, c in enumerate(expect))
        out["fasta_identical_to_resident_batch"] = (want == text)
    return out


def end_to_end_long(piles, jobs=10, parallel=3, repeats=E2E_REPEATS, settings=None, server_cmd=None, job_cmd=None):
    """A stream long enough to call a rate a rate: `jobs` jobs of the whole text (each `repeats` x the step's piles:
    30 720 of them, 25 GB) through ONE consensus server, `parallel` of them at a time -- the devices' queues never run
    dry between jobs, so what the server sees is one stream of jobs x 30 720 piles (the file itself cannot be made a
    hundred times longer: it would not fit the scratch disk, and a pipe delivers a tenth of the rate).  The clock runs
    from the first job's start to the last job's end; server start is not on it (a node pays it once).
    `settings`: [(name, {environment of the server})], e.g. the worker's batch size -- one server per setting, one
    after the other, same text.  Every job's FASTA must be the same bytes."""
    import hashlib
    import shutil
    import signal
    settings = settings or [("batch 0.4 G bases (the default)", {}),
                            ("batches that grow: x 1.5 from the 4th on, up to 1.4 G bases; 4 text buffers ahead",
                             {"FALCON_AMD_BATCH_GROW": "3:1.5:1400000000", "FALCON_AMD_READ_AHEAD": "4"}),
                            ("batch 1.3 G bases, 4 text buffers ahead", {"FALCON_AMD_BATCH_BASES": "1300000000",
                                                                         "FALCON_AMD_READ_AHEAD": "4"})]
    out = {"settings": []}
    with tempfile.TemporaryDirectory() as tmp:
        src, sock = os.path.join(tmp, "piles.txt"), os.path.join(tmp, "srv.sock")
        per_repeat = sum(sum(len(x) + 10 for x in p) for p in piles) + 1
        repeats = max(1, min(repeats, int(shutil.disk_usage(tmp).free * 0.4 // per_repeat)))
        with open(src, "wb") as f:
            write_la4falcon(piles, f, repeats)
        size = os.path.getsize(src)
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))}
        # (`server_cmd` / `job_cmd`: the CPU tests' stand-ins)
        cmd = job_cmd or [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt", "0.70",
                          "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"]
        n_job = repeats * len(piles)
        digest0 = None
        for name, senv in settings:
            srv = subprocess.Popen((server_cmd or [sys.executable, "-m", "falcon_amd.mains.consensus_server"]) +
                                   ["--socket", sock], cwd=ROOT, env=dict(env, **senv), stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, text=True)
            row = {"setting": name}
            try:
                line = srv.stdout.readline()
                if "ready" not in line:
                    raise RuntimeError("the consensus server did not come up: %s %s" % (line, srv.stderr.read()[-400:]))
                running, done_at, started, digests = [], [], 0, []
                t0 = time.perf_counter()
                while started < jobs or running:
                    while started < jobs and len(running) < parallel:
                        dst = os.path.join(tmp, "cns_%d.fasta" % started)
                        fin, fout = open(src), open(dst, "w")
                        running.append((subprocess.Popen(cmd, stdin=fin, stdout=fout, cwd=ROOT,
                                                         env=dict(env, FALCON_AMD_SERVER=sock)), fin, fout, dst,
                                        time.perf_counter()))
                        started += 1
                    for item in list(running):
                        p, fin, fout, dst, t_job = item
                        if p.poll() is None:
                            continue
                        running.remove(item)
                        fin.close(); fout.close()
                        if p.returncode != 0:
                            raise RuntimeError("end_to_end_long: a job exited %d" % p.returncode)
                        done_at.append((time.perf_counter() - t0, time.perf_counter() - t_job))
                        h = hashlib.sha1()
                        with open(dst, "rb") as f:
                            for blk in iter(lambda: f.read(1 << 24), b""):
                                h.update(blk)
                        digests.append(h.hexdigest()[:16])
                        os.unlink(dst)
                    if time.perf_counter() - t0 > 1500:
                        raise RuntimeError("end_to_end_long: the jobs did not finish")
                    time.sleep(0.002)
                wall = max(t for t, _ in done_at)
                digest0 = digest0 or digests[0]
                # the rate between the first job's end and the last job's end: start-up and the first jobs' ramp excluded
                firsts = sorted(t for t, _ in done_at)
                steady = (len(firsts) - 1) * n_job / (firsts[-1] - firsts[0]) if len(firsts) > 1 and firsts[-1] > firsts[0] else None
                row.update({"piles": jobs * n_job, "wall_s": round(wall, 2), "piles_per_sec": round(jobs * n_job / wall, 1),
                            "text_GB_per_sec": round(jobs * size / 1e9 / wall, 2),
                            "piles_per_sec_between_first_and_last_job_end": round(steady, 1) if steady else None,
                            "job_wall_s": [round(w, 2) for _, w in done_at],
                            "every_fasta_identical": len(set(digests)) == 1 and digests[0] == digest0,
                            "fasta_sha1": digests[0]})
            finally:
                srv.send_signal(signal.SIGTERM)
                try:
                    srv.wait(timeout=120)
                except subprocess.TimeoutExpired:
                    srv.kill()
            out["settings"].append(row)
    out["what"] = ("%d jobs of %d piles (%.0f MB of text each, from the page cache), %d at a time, through one consensus "
                   "server per setting (started before, not on the clock): first job's start to last job's end"
                   % (jobs, n_job, size / 1e6, parallel))
    return out
