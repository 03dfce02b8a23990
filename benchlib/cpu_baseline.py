"""bench.py, the `cpu_baseline` leg: the reference's own C (oracle/_ref, built by oracle/Makefile from
/root/reference/src/c) -- or, where that build is absent, the repo's restatement (oracle/) -- timed in worker processes on the
host's cores over a bounded sample of the bench's piles.  TEST / MEASUREMENT INFRASTRUCTURE: this is the one place outside
tests/ and __graft_entry__.smoke() that loads anything under oracle/, and only as the checker and the reported baseline --
never as the thing measured as `value`, never by the product."""
from __future__ import annotations

import multiprocessing as mp
import os
import resource
import time

from benchlib.workloads import K, MIN_COV, MIN_IDT


# ---- CPU baseline workers (own processes: the reference C is not re-entrant,
# ---- falcon.c:338, and pays a one-off 0.9 GB workspace per process) ----------
def _cpu_worker(args):
    kind, piles, cpu = args
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})  # one physical core each, spread over the L3 domains (see _placement)
        except OSError:
            pass
    from oracle.pyoracle import Port, Ref
    impl = Ref() if kind == "reference" else Port()
    impl.generate_consensus(piles[0], MIN_COV, K, MIN_IDT)  # warm-up, untimed
    r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
    out = [impl.generate_consensus(p, MIN_COV, K, MIN_IDT)[0] for p in piles[1:]]
    dt, r1 = time.perf_counter() - t0, resource.getrusage(resource.RUSAGE_SELF)
    # where the time went: user / kernel seconds and page faults of the timed piles (the reference
    # calloc()s ~200 MB per alignment, DW_banded.c:164, and sweeps a 0.9 GB workspace per pile,
    # falcon.c:293-298: page zeroing and faults are kernel time, and they serialise on a big host)
    use = {"user_s": r1.ru_utime - r0.ru_utime, "sys_s": r1.ru_stime - r0.ru_stime,
           "minor_faults": r1.ru_minflt - r0.ru_minflt, "major_faults": r1.ru_majflt - r0.ru_majflt}
    return out, dt, use


def host_cores():
    """(logical cpus this process may use, physical cores of the host or None)."""
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    try:
        import psutil
        phys = psutil.cpu_count(logical=False)
    except Exception:
        phys = None
    return logical, phys


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable:"):
                    return int(ln.split()[1]) / 1048576.0
    except OSError:
        pass
    return None


def _cgroup_cpu():
    """(cpus the container's CPU quota allows or None, {throttled periods, throttled microseconds} so far)."""
    quota, stat = None, {}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                w = f.read().split()
            if path.endswith("cpu.max"):
                quota = None if w[0] == "max" else float(w[0]) / float(w[1])
            else:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                    quota = None if int(w[0]) < 0 else float(w[0]) / float(g.read())
            break
        except (OSError, ValueError, IndexError):
            continue
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            with open(path) as f:
                for ln in f:
                    k, v = ln.split()
                    if k in ("nr_throttled", "throttled_usec", "throttled_time", "nr_periods"):
                        stat[k] = int(v)
            break
        except (OSError, ValueError):
            continue
    return quota, stat


def _placement():
    """The cpus workers are pinned to, in the order they are handed out: one logical cpu per
    physical core, the cores dealt round robin over the L3 domains (a CCD of an EPYC) and with them
    over the sockets -- 8 workers sit on 8 different L3s, 16 on 16 ....  [] when the topology
    cannot be read (then nobody is pinned)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return []
    seen_core, by_l3 = set(), {}
    for c in allowed:
        base = "/sys/devices/system/cpu/cpu%d/" % c
        try:
            with open(base + "topology/physical_package_id") as f:
                pkg = int(f.read())
            with open(base + "topology/core_id") as f:
                core = int(f.read())
            try:
                with open(base + "cache/index3/id") as f:
                    l3 = int(f.read())
            except OSError:
                l3 = pkg
        except (OSError, ValueError):
            return []
        if (pkg, core) in seen_core:
            continue  # (the SMT sibling of a core already taken)
        seen_core.add((pkg, core))
        by_l3.setdefault((pkg, l3), []).append(c)
    order, k = [], 0
    groups = [by_l3[g] for g in sorted(by_l3)]
    while any(k < len(g) for g in groups):
        order += [g[k] for g in groups if k < len(g)]
        k += 1
    return order


def _cpu_run(kind, piles, first, cores, per, cpus):
    """`cores` worker processes, worker w on piles[first + w * per : first + (w + 1) * per]
    (its first pile is the untimed warm-up), pinned to cpus[w] when there is one.
    -> (result object, {pile index: string})"""
    jobs = [(kind, piles[first + i * per:first + (i + 1) * per], cpus[i] if i < len(cpus) else None)
            for i in range(cores)]
    ctx = mp.get_context("fork")
    _q, thr0 = _cgroup_cpu()
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    _q, thr1 = _cgroup_cpu()
    strings = {}
    for w, (out, _, _u) in enumerate(res):
        for j, s in enumerate(out):
            strings[first + w * per + 1 + j] = s
    bases = sum(len(s) for s in strings.values())
    busy = max(r[1] for r in res)  # workers run concurrently: timed span of the slowest
    n = max(1, len(strings))
    use = {k: sum(r[2][k] for r in res) for k in ("user_s", "sys_s", "minor_faults", "major_faults")}
    return {"cores": cores, "value": round(bases / busy, 1), "piles_per_sec": round(len(strings) / busy, 3),
            "per_core_bases_per_sec": round(bases / sum(r[1] for r in res), 1), "piles": len(strings),
            "wall_s": round(wall, 1), "pinned": bool(cpus),
            # the container's CPU quota at work: scheduler periods in which the workers were stopped, and for how long
            "cgroup_throttled": {k: thr1.get(k, 0) - thr0.get(k, 0) for k in thr1},
            # per timed pile: seconds in user code / in the kernel, page faults
            "per_pile": {"user_s": round(use["user_s"] / n, 3), "sys_s": round(use["sys_s"] / n, 3),
                         "minor_faults": int(use["minor_faults"] / n), "major_faults": int(use["major_faults"] / n)}}, strings


def cpu_baseline(piles, timed_per_worker=12, procs=None):
    """-> (the cpu_baseline object, {pile index: consensus string} of the timed piles).

    SURVEY.md 8d "CPU baseline timing": the reference C path in worker processes, one untimed
    warm-up pile per worker (the 0.9 GB workspace), `timed_per_worker` timed piles each, taken from
    the front of this rank's batch.  How many processes the host runs best is MEASURED: 1 (the
    per-core figure), 8, 16, 24, 32, 64 and the physical cores (never more than this process may
    run on, nor than memory allows), every worker pinned to a physical core of its own, the cores
    dealt over the L3 domains; `value` is the best of them, all are listed -- each with the
    seconds its piles spent in user code and in the kernel and their page faults, which is what
    the reference's anti-scaling on a big host is made of (every alignment calloc()s ~200 MB,
    DW_banded.c:164; every pile sweeps a 0.9 GB workspace, falcon.c:293-298)."""
    from oracle.pyoracle import build, have_ref
    try:
        build()
    except Exception:
        pass
    kind = "reference" if have_ref() else "port"
    logical, phys = host_cores()
    cores = max(1, min(logical, phys or logical))
    mem = _mem_available_gb()
    if mem is not None:
        cores = max(1, min(cores, int(mem / 2.0)))  # 0.9 GB workspace + the piles + headroom
    cpus = _placement()
    quota, _thr = _cgroup_cpu()
    runs, strings, first = [], {}, 0
    # (the larger configurations time fewer piles per worker: every configuration ~5-20 s)
    wanted = sorted({min(cores, int(x)) for x in procs} if procs else
                    {1, min(cores, 8), min(cores, 16), min(cores, 24), min(cores, 32), min(cores, 64), cores})
    for want in wanted:
        per = min((timed_per_worker if want <= 32 else max(4, timed_per_worker // 3)) + 1, len(piles))
        c = max(1, min(want, (len(piles) - first) // per))
        if (c < want and runs) or first + per > len(piles):
            break  # (not enough piles left for another configuration)
        r, got = _cpu_run(kind, piles, first, c, per, cpus)
        runs.append(r)
        strings.update(got)
        first += c * per
    best = max(runs, key=lambda r: r["value"])
    one = next((r for r in runs if r["cores"] == 1), None)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": best["value"], "unit": "bases/s", "cores": best["cores"], "kind": kind,
        "piles_per_sec": best["piles_per_sec"],
        # the per-core figure: ONE process alone on the host (no neighbour sweeping its workspace)
        "per_core_bases_per_sec": (one or best)["per_core_bases_per_sec"],
        "host_cpu_count": logical, "host_physical_cores": phys, "host_cpu_model": cpu_model,
        "workers_pinned_to_cores_over_l3_domains": bool(cpus),
        # what the box lets this process group use at all: a container with a CPU quota runs 128
        # workers no faster than `quota` of them (runs[].cgroup_throttled shows it happening)
        "cgroup_cpu_quota_cores": quota,
        "runs": runs,
        "sample": "piles of this workload from the front of the batch, %d timed per worker process (%d "
                  "beyond 32 processes) + 1 untimed warm-up pile each; worker processes: %s (capped by the "
                  "cpus allowed and memory / 2 GB), each pinned to a physical core of its own -- `value` "
                  "is the best of them: %d processes, %d piles; `runs[].per_pile` = user / kernel "
                  "seconds and page faults per timed pile"
                  % (timed_per_worker, max(4, timed_per_worker // 3), ", ".join(str(r["cores"]) for r in runs),
                     best["cores"], best["piles"]),
    }, strings
