#!/usr/bin/env python3
"""bench.py -- FALCON pre-assembly consensus hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (seed k-mer index -> k-mer chaining ->
banded O(ND) alignment + trace-back -> MSA sweep + consensus back-trace) over one
resident batch of synthetic LA4Falcon-style piles, BASELINE.json configs[1]:
E. coli-like ~20 kb seeds x 40x coverage, flags of examples/fc_run_ecoli.cfg:33
(--min-cov 4 --min-idt 0.70 --max-n-read 200).  Inputs are 2-bit packed and resident
in HBM before the timed region starts; every rank owns its own piles (piles are
independent: no collective on the data path, weak scaling).

Prints ONE JSON line (rank 0): metric = consensus bases/s over all GPUs, plus
  roofline     -- dominant kernel, algorithmic bytes / HIP-event time
  cpu_baseline -- the reference C path (oracle/_ref, else our restatement) timed on the
                  host cores on a bounded sample of the same piles (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md

MIN_COV, K, MIN_IDT, MAX_N_READ = 4, 8, 0.70, 200


def _gen_pile(seed):
    from falcon_amd.synth import codes_to_str, make_pile, pile_to_seqs
    s, rd = make_pile(seed, S=20000, coverage=40.0)
    return [codes_to_str(x).encode("ascii") for x in pile_to_seqs(s, rd, MAX_N_READ)]


def gen_piles(seeds, procs):
    if procs <= 1 or len(seeds) < 4:
        return [_gen_pile(s) for s in seeds]
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        return pool.map(_gen_pile, seeds, chunksize=4)


# ---- CPU baseline workers (own processes: the reference C is not re-entrant,
# ---- falcon.c:338, and pays a one-off 0.9 GB workspace per process) ----------
def _cpu_worker(args):
    kind, piles = args
    from oracle.pyoracle import Port, Ref
    impl = Ref() if kind == "reference" else Port()
    impl.generate_consensus(piles[0], MIN_COV, K, MIN_IDT)  # warm-up, untimed
    t0 = time.perf_counter()
    bases = 0
    for p in piles[1:]:
        bases += len(impl.generate_consensus(p, MIN_COV, K, MIN_IDT)[0])
    return bases, len(piles) - 1, time.perf_counter() - t0


def cpu_baseline(piles):
    from oracle.pyoracle import build, have_ref
    try:
        build()
    except Exception:
        pass
    kind = "reference" if have_ref() else "port"
    cores = max(1, min(os.cpu_count() or 1, 16, len(piles) // 3))
    per = max(2, min(9, len(piles) // cores))  # 1 warm-up + up to 8 timed piles per worker
    jobs = [(kind, piles[i * per:(i + 1) * per]) for i in range(cores)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    bases = sum(r[0] for r in res)
    n = sum(r[1] for r in res)
    busy = max(r[2] for r in res)  # workers run concurrently: timed span of the slowest
    return {
        "value": round(bases / busy, 1), "unit": "bases/s", "cores": cores, "kind": kind,
        "piles_per_sec": round(n / busy, 3),
        "per_core_bases_per_sec": round(bases / sum(r[2] for r in res), 1),
        "sample": "%d piles of this workload (+1 untimed warm-up pile per worker process), "
                  "%d worker processes, %.1f s wall incl. warm-up" % (n, cores, wall),
    }


def write_la4falcon(piles, f):
    """The LA4Falcon text a pile [seed, seed copy + reads by length] (falcon_amd.synth
    pile_to_seqs) came from: the seed line, then the reads (the reader adds the seed's copy
    itself, consensus.py:183-190)."""
    for i, p in enumerate(piles):
        lines, seen_copy = [b"%09d %s" % (i, p[0])], False
        for j, r in enumerate(p[1:]):
            if not seen_copy and r == p[0]:
                seen_copy = True
                continue
            lines.append(b"%09d %s" % (1000000 + 1000 * i + j, r))
        f.write(b"\n".join(lines) + b"\n+ +\n")
    f.write(b"- -\n")


def end_to_end(piles, extra_args=()):
    """SURVEY.md 8d "end-to-end": LA4Falcon text on stdin -> FASTA on stdout through the
    consensus worker (falcon_amd.mains.consensus: native reader, staging, GPU stages,
    printing) in a process of its own, on the piles of this workload written out as text.
    Reported beside `value` (kernel-only, inputs resident in HBM), never as it."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "piles.txt"), os.path.join(tmp, "cns.fasta")
        with open(src, "wb") as f:
            write_la4falcon(piles, f)
        size = os.path.getsize(src)
        cmd = [sys.executable, "-m", "falcon_amd.mains.consensus", "--output-multi", "--min-idt",
               "0.70", "--min-cov", "4", "--max-n-read", "200", "--n-core", "1"] + list(extra_args)
        t0 = time.perf_counter()
        with open(src) as fin, open(dst, "w") as fout:
            # (a profiler wrapped around this process stays with this process: the worker's
            # launches are at another batch size and would blur its per-kernel averages)
            env = {k: v for k, v in os.environ.items()
                   if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "ROCTRACER", "HSA_TOOLS"))}
            subprocess.run(cmd, stdin=fin, stdout=fout, check=True, cwd=root, timeout=600, env=env)
        wall = time.perf_counter() - t0
        with open(dst) as f:
            bases = sum(len(ln) - 1 for ln in f if not ln.startswith(">"))
    return {"piles_per_sec": round(len(piles) / wall, 1), "text_MB_per_sec": round(size / 1e6 / wall, 1),
            "fasta_bases_per_sec": round(bases / wall, 1), "wall_s": round(wall, 2),
            "what": "%d piles (%.0f MB of text from the page cache) -> FASTA, one worker process on "
                    "one GPU, process start and HIP initialisation included" % (len(piles), size / 1e6)}


def measured_stream_rate(torch, mib=1024, reps=5):
    """GB/s (read + write) of a device-to-device copy of `mib` MiB: what HBM delivers to the
    simplest streaming kernel on this box (outside the timed region; plumbing, not product)."""
    try:
        a = torch.empty(mib << 20, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del a, b
        return round(2.0 * (mib << 20) / (ms * 1e-3) / 1e9, 1)
    except Exception:  # informative only
        return None


def measured_traffic(kernel, piles):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes committed under
    profiles/ (FETCH/WRITE counters cannot be read from inside this process); None unless
    a measurement of this kernel at this batch size is on file."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(kernel)
        if rec and int(rec["piles_per_launch"]) == int(piles):
            return int(rec["hbm_bytes_per_launch"]), rec.get("what", "")
    except Exception:
        pass
    return None, "no PMC measurement of this kernel at this batch size under profiles/"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--piles", type=int, default=int(os.environ.get("FALCON_BENCH_PILES", "3072")),
                    help="piles per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # synthetic input first (forks worker processes; no GPU state exists yet)
    ncpu = os.cpu_count() or 1
    procs = max(1, min(32, ncpu // max(1, world)))
    seeds = [1000003 * (rank + 1) + i for i in range(args.piles)]
    t_gen = time.perf_counter()
    piles = gen_piles(seeds, procs)
    t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from falcon_amd.engine import Engine

    eng = Engine(local_rank)
    t_up = time.perf_counter()
    batch = eng.batch(piles)  # ASCII -> HBM, packed to 2 bits/base on the GPU
    t_up = time.perf_counter() - t_up

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        batch.run(MIN_COV, K, MIN_IDT)
    sync()
    t0 = time.perf_counter()
    ms_align = ms_cns = ms_chain = ms_index = ms_total = 0.0
    ms_k = {"k_tags": 0.0, "k_links": 0.0, "k_score": 0.0, "k_backtrace": 0.0}
    for _ in range(args.steps):
        batch.run(MIN_COV, K, MIN_IDT)  # returns after the stream drained
        st = batch.stats()
        ms_align += st.ms_align
        ms_cns += st.ms_consensus
        ms_chain += st.ms_chain
        ms_index += st.ms_index
        ms_total += st.ms_total
        ms_k["k_tags"] += st.ms_tags          # k_tags + k_tscan (0.7 ms)
        ms_k["k_links"] += st.ms_links
        ms_k["k_score"] += st.ms_score
        ms_k["k_backtrace"] += st.ms_backtrace
    sync()
    elapsed = time.perf_counter() - t0

    stream_gbs = measured_stream_rate(torch)

    st = batch.stats()
    # whole-job aggregate: units summed over ranks, time = slowest rank
    from falcon_amd.multigpu import reduce_measurement
    bases_all, piles_all, elapsed = reduce_measurement(float(st.O), float(st.n_piles), elapsed,
                                                       device="cuda")

    if rank == 0:
        k = max(1, args.steps)
        stage_ms = {"index": ms_index / k, "chain": ms_chain / k, "align": ms_align / k,
                    "consensus": ms_cns / k}
        # between k_align and the MSA kernels the host sizes the MSA pools from the
        # alignment summaries (D2H, O(#reads) loop, H2D): device-idle time of the step
        host_gap = ms_total / k - sum(stage_ms.values())
        # algorithmic bytes per launch
        alg = {
            "index": st.T // 4 + 8 * st.T + 2 * 4 * 65537 * st.n_piles,
            "chain": st.L // 4,
            "align": st.L // 4 + 4 * st.C + 8 * st.D,
            "consensus": 16 * st.A + 12 * st.T + 5 * st.O,  # k_tags + k_links + k_score + k_backtrace
        }
        # per kernel, HIP events on the engine's stream around every launch of the timed
        # region (the index / chain / align stages are one kernel each)
        kernel_ms = {"k_seed_index": stage_ms["index"], "k_chain": stage_ms["chain"],
                     "k_align": stage_ms["align"]}
        kernel_ms.update({n: v / k for n, v in ms_k.items()})
        kalg = {"k_seed_index": alg["index"], "k_chain": alg["chain"], "k_align": alg["align"],
                "k_tags": 8 * st.D + 4 * st.A,            # script in, one tag word per column out
                "k_links": 4 * st.A + 12 * st.T,          # tag words in, per-position arrays
                "k_score": 8 * st.A,                      # link words in, node records out (upper bound)
                "k_backtrace": 5 * st.O}
        domk = max(kernel_ms, key=lambda n: kernel_ms[n])
        ach = kalg[domk] / (kernel_ms[domk] * 1e-3) / 1e9 if kernel_ms[domk] > 0 else 0.0
        traffic, traffic_src = measured_traffic(domk, args.piles)
        out = {
            "metric": "consensus_bases_per_sec",
            "value": round(bases_all * args.steps / elapsed, 1),
            "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (2-bit packed bases, integer DP)",
            "data": "synthetic",
            "config": {
                "workload": "E. coli-like piles: ~20 kb seed x 40x coverage, e=0.13 "
                            "(BASELINE.json configs[1]; falcon_amd/synth.py, SURVEY.md 8d)",
                "piles_per_step_per_gpu": args.piles,
                "flags": "--min-cov 4 --min-idt 0.70 --max-n-read 200 (K=8)",
                "sequences_per_step_per_gpu": int(st.n_seqs),
                "accepted_alignments_per_step_per_gpu": int(st.n_aligned),
            },
            "piles_per_sec": round(piles_all * args.steps / elapsed, 2),
            "roofline": {
                "bound": "hbm", "kernel": domk,
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5),
                # the peak a plain device copy reaches on this box, and the
                # fraction against that denominator too
                "peak_measured_copy": stream_gbs,
                "frac_of_measured": round(ach / stream_gbs, 5) if stream_gbs else None,
                "algorithmic_bytes_per_launch": int(kalg[domk]),
                "avg_launch_ms": round(kernel_ms[domk], 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
            },
            "stage_ms": {s: round(v, 4) for s, v in stage_ms.items()},
            "kernel_ms": {n: round(v, 4) for n, v in kernel_ms.items()},
            "host_plan_gap_ms": round(host_gap, 3),
            "path_b_alg_bytes_per_step": int(st.b_alg()),
            "path_frac_of_hbm_roofline": round(
                st.b_alg() * world * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "setup_s": {"generate": round(t_gen, 2), "stage_to_hbm_incl_pcie": round(t_up, 2),
                        "stage_to_hbm_incl_pcie_again": None},
        }
        if world == 1 and not args.no_end_to_end:
            try:
                out["end_to_end"] = end_to_end(piles)
            except Exception as e:  # informative; never lose the GPU line
                out["end_to_end"] = {"piles_per_sec": None, "what": "failed: %r" % (e,)}
        if world == 1:
            # staging again, now that the context's pinned and device staging buffers exist:
            # the steady-state cost of handing a batch of host buffers over (outside `value`;
            # after the end-to-end leg: amdgpu wipes released VRAM and a worker process
            # starting right behind a large release waits for that)
            try:
                t_up2 = time.perf_counter()
                again = eng.batch(piles)
                t_up2 = time.perf_counter() - t_up2
                again.free()
                out["setup_s"]["stage_to_hbm_incl_pcie_again"] = round(t_up2, 2)
            except Exception:  # informative; never lose the GPU line
                pass
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(piles[:144])
            except Exception as e:  # the baseline is informative; never lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "bases/s", "cores": 0,
                                       "kind": "unavailable", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    batch.free()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
