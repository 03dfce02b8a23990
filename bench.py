#!/usr/bin/env python3
"""bench.py -- FALCON pre-assembly consensus hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ecoli|dmel|arab]

A "step" is one pass of the whole hot path (seed k-mer index -> k-mer chaining ->
banded O(ND) alignment + trace-back -> MSA sweep + consensus back-trace) over one
resident batch of synthetic LA4Falcon-style piles.  Default workload = BASELINE.json
configs[1]: E. coli-like ~20 kb seeds x 40x coverage, flags of
examples/fc_run_ecoli.cfg:33 (--min-cov 4 --min-idt 0.70 --max-n-read 200); `dmel`
(configs[3], 30 kb x 80x, the 200-read cap binding) and `arab` (configs[4], 25 kb x 60x,
two haplotypes 0.5 % apart) are SURVEY.md 8d's other two.  Inputs are 2-bit packed and
resident in HBM before the timed region starts; every rank owns its own piles (piles are
independent: no collective on the data path, weak scaling).

`--gpus N` with N > 1 starts N ranks itself (one process per GPU under
torch.distributed.run, 127.0.0.1 rendezvous) unless a launcher already did
(WORLD_SIZE set); the line is only printed when the ranks that ran == N.

Prints ONE JSON line (rank 0): metric = consensus bases/s over all GPUs, plus
  roofline     -- dominant kernel, algorithmic bytes / HIP-event time
  cpu_baseline -- the reference C path (oracle/_ref, else our restatement) timed on the
                  host cores on a bounded sample of the same piles (rank 0, N=1 only), and
  parity_*     -- the GPU's consensus of exactly those sample piles compared with the
                  strings the CPU baseline just produced (outside the timed region).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import multiprocessing as mp
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (the parts of this file live in benchlib/: the input, the CPU baseline, the end-to-end legs, the
# HBM traffic on file; the names stay importable from here)
from benchlib.cpu_baseline import cpu_baseline, host_cores  # noqa: E402,F401
from benchlib.e2e import (E2E_REPEATS, end_to_end, end_to_end_multi, end_to_end_served,  # noqa: E402,F401
                          end_to_end_workers)
from benchlib.traffic import (KERNEL_NAME, _code_only, kernel_source_sha, measured_issue,  # noqa: E402,F401
                              measured_stream_rate, measured_traffic)
from benchlib.workloads import (HBM_PEAK_GBS, K, MAX_N_READ, MIN_COV, MIN_IDT, WORKLOADS, _gen_pile,  # noqa: E402,F401
                                gen_piles, write_la4falcon)


                  # that start-up is amortised the way a .las block's tens of thousands amortise it


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["trim", "align1500", "utg", "e2e-long"], default="ecoli",
                    help="ecoli / dmel / arab: BASELINE.json's configs of the falcon_sense path; trim / align1500 / utg: "
                         "the SURVEY.md 8(f) paths beside it (benchlib/secondary.py), one GPU, their own metrics")
    ap.add_argument("--piles", type=int, default=int(os.environ.get("FALCON_BENCH_PILES", "0")),
                    help="piles per step per GPU (default: 3072 ecoli, 1024 dmel, 1536 arab)")
    ap.add_argument("--jobs", type=int, default=10, help="e2e-long: jobs of the whole text through one server")
    ap.add_argument("--parallel", type=int, default=3, help="e2e-long: jobs at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-procs", default="",
                    help="worker-process counts of the CPU baseline, comma separated (default: 1,8,16,24,32,64 and "
                         "the physical cores)")
    ap.add_argument("--cpu-baseline-timed", type=int, default=12,
                    help="timed piles per CPU worker process (every one is also a parity check of the GPU's answer)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="resident batches taking turns in the pipelined steps (>= 2): with 3, a "
                         "batch's front stages are queued while the one before it is aligned and "
                         "the one before that goes through its MSA stage")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one resident batch, every step drained before the next (A/B of the "
                         "two-batch pipelining)")
    args = ap.parse_args(argv)
    args.piles_given = args.piles > 0
    if args.piles <= 0 and args.workload in WORKLOADS:
        args.piles = WORKLOADS[args.workload]["piles"]
    return args


def relaunch_under_torchrun(args, argv):
    """`bench.py --gpus N` started bare: become N ranks (one per GPU, RCCL) the way the
    driver would have started them.  Fails loudly when the box has fewer devices."""
    from falcon_amd.lib import load
    n_dev = load().fa_device_count()
    if os.environ.get("FALCON_BENCH_ONE_DEVICE"):
        n_dev = max(n_dev, args.gpus) if n_dev > 0 else 0   # (tests: every rank on device 0, see Plumbing)
    if n_dev < args.gpus:
        sys.exit("bench.py: --gpus %d asked for, %d HIP device(s) visible -- refusing to run "
                 "fewer ranks than asked for" % (args.gpus, n_dev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    os.execv(sys.executable, cmd)


class Plumbing:
    """Device/rendezvous plumbing of a rank: torch.distributed over RCCL ("nccl") on the GPU
    box; the CPU tests run the same rank logic over gloo with a stand-in engine.
    FALCON_BENCH_BACKEND=gloo + FALCON_BENCH_ONE_DEVICE=1 (tests/test_gpu_cli.py): the REAL engine
    under world > 1 on a box with one GPU -- every rank on device 0 (RCCL refuses two ranks on
    one device, gloo does not care), the ranks lined up and the measurement reduced on host
    tensors.  What stays RCCL-only is the transport of those three numbers."""

    def __init__(self, rank, local_rank, world, backend="nccl"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        backend = os.environ.get("FALCON_BENCH_BACKEND") or backend
        self.rank, self.local_rank, self.world, self.backend = rank, local_rank, world, backend
        self.cuda = backend == "nccl"                       # collectives on device tensors
        self.device = 0 if os.environ.get("FALCON_BENCH_ONE_DEVICE") else local_rank
        self.gpu = self.cuda or bool(os.environ.get("FALCON_BENCH_ONE_DEVICE"))   # a HIP device behind the engine
        if self.gpu:
            torch.cuda.set_device(self.device)
        if world > 1:
            kw = {"device_id": torch.device("cuda", local_rank)} if self.cuda else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
            if dist.get_world_size() != world:
                raise RuntimeError("rendezvous saw %d ranks, %d expected" % (dist.get_world_size(), world))

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def bench_rank(args, plumb, make_engine, piles, t_gen=0.0, out=None):
    """What one rank does with its piles.  Returns the JSON object on rank 0 (also printed),
    else None."""
    rank, world = plumb.rank, plumb.world
    wl = WORKLOADS[args.workload]
    out = sys.stdout if out is None else out

    eng = make_engine(getattr(plumb, "device", plumb.local_rank))
    t_up = time.perf_counter()
    batch = eng.batch(piles)  # ASCII -> 2 bits/base on the host -> HBM
    t_up = time.perf_counter() - t_up
    # Steps are pipelined the way a streaming job runs them: `--in-flight` resident batches
    # (the same piles staged that many times) take turns, two submits ahead of every wait, so
    # that a step's index, chaining and alignment run beside the consensus stage (tags, links,
    # score recurrence, back-trace) of the step before -- fa_batch_submit / fa_batch_wait.
    # Every step is still one full pass of the whole path over one batch.
    pipelined = hasattr(batch, "submit") and not args.no_pipeline
    depth = max(2, args.in_flight)
    pair = [batch] + [eng.batch(piles) for _ in range(depth - 1)] if pipelined else [batch]
    for b in pair:  # (first run of a batch sizes its MSA buffers: setup, not a step)
        b.run(MIN_COV, K, MIN_IDT)

    acc = {}
    submit_ms = []  # host time of every fa_batch_submit of the timed region

    def steps(n, record):
        if n <= 0:
            return
        if not pipelined:
            for _ in range(n):
                batch.run(MIN_COV, K, MIN_IDT)  # returns after the streams drained
                if record:
                    tally(batch.stats())
            return
        # step i runs on batch i mod depth; depth - 1 submits stay ahead of every wait
        def submit(bt):
            t_s = time.perf_counter()
            bt.submit(MIN_COV, K, MIN_IDT)
            if record:
                submit_ms.append((time.perf_counter() - t_s) * 1e3)

        for j in range(min(depth - 1, n)):
            submit(pair[j % depth])
        for i in range(n):
            if i + depth - 1 < n:
                submit(pair[(i + depth - 1) % depth])
            cur = pair[i % depth]
            cur.wait()
            if record:
                tally(cur.stats())

    def tally(st):
        for n in ("ms_align", "ms_consensus", "ms_chain", "ms_index", "ms_total", "ms_tags",
                  "ms_links", "ms_score", "ms_backtrace"):
            acc[n] = acc.get(n, 0.0) + getattr(st, n)

    steps(args.warmup, False)
    plumb.sync()
    t0 = time.perf_counter()
    steps(args.steps, True)
    plumb.sync()
    elapsed = time.perf_counter() - t0

    stream_gbs = measured_stream_rate(plumb.torch) if getattr(plumb, "gpu", plumb.cuda) else None

    # the dominant kernel with the device to itself (outside the timed region): in the
    # pipelined steps its launches share the CUs with the previous step's consensus stage
    alone_ms = {}
    if pipelined and rank == 0:
        for _ in range(2):
            batch.run(MIN_COV, K, MIN_IDT)
        s1 = batch.stats()
        alone_ms = {"k_seed_index": s1.ms_index, "k_chain": s1.ms_chain, "k_align": s1.ms_align,
                    "k_tags": s1.ms_tags, "k_links": s1.ms_links, "k_score": s1.ms_score,
                    "k_backtrace": s1.ms_backtrace}

    st = batch.stats()
    # whole-job aggregate: units summed over ranks, time = slowest rank
    from falcon_amd.multigpu import reduce_measurement
    own_elapsed = elapsed
    bases_all, piles_all, elapsed = reduce_measurement(float(st.O), float(st.n_piles), elapsed,
                                                       device="cuda" if plumb.cuda else None)
    from falcon_amd.multigpu import gather_per_rank
    per_rank = gather_per_rank([float(st.O) * args.steps / own_elapsed, own_elapsed / max(1, args.steps) * 1e3,
                                acc.get("ms_align", 0.0) / max(1, args.steps)],
                               device="cuda" if plumb.cuda else None)
    ranks_ran = world
    if world > 1:
        one = plumb.torch.ones(1, dtype=plumb.torch.float64, device="cuda" if plumb.cuda else None)
        plumb.dist.all_reduce(one)
        ranks_ran = int(one.item())
    res = None
    if rank == 0:
        if ranks_ran != args.gpus:
            raise RuntimeError("bench.py --gpus %d: %d rank(s) took part" % (args.gpus, ranks_ran))
        k = max(1, args.steps)
        stage_ms = {"index": acc["ms_index"] / k, "chain": acc["ms_chain"] / k,
                    "align": acc["ms_align"] / k, "consensus": acc["ms_consensus"] / k}
        # between the alignment and the MSA kernels the host (the context's planner thread)
        # sizes the MSA pools from the alignment summaries (D2H, O(#reads) loop, H2D): a gap in
        # THIS batch's kernels, during which the device runs the next batch's (pipelined
        # steps: the figure also holds the MSA kernels' wait for wave slots)
        host_gap = acc["ms_total"] / k - sum(stage_ms.values())
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
                     "k_align": stage_ms["align"],
                     "k_tags": acc["ms_tags"] / k,          # k_tags + k_tscan (0.7 ms)
                     "k_links": acc["ms_links"] / k, "k_score": acc["ms_score"] / k,
                     "k_backtrace": acc["ms_backtrace"] / k}
        kalg = {"k_seed_index": alg["index"], "k_chain": alg["chain"], "k_align": alg["align"],
                "k_tags": 8 * st.D + 4 * st.A,            # script in, one tag word per column out
                "k_links": 4 * st.A + 12 * st.T,          # tag words in, per-position arrays
                "k_score": 8 * st.A,                      # link words in, node records out (upper bound)
                "k_backtrace": 5 * st.O}
        domk = max(kernel_ms, key=lambda n: kernel_ms[n])
        ach = kalg[domk] / (kernel_ms[domk] * 1e-3) / 1e9 if kernel_ms[domk] > 0 else 0.0
        traffic, traffic_src = measured_traffic(domk, args.piles, args.workload)
        if domk == "k_align" and os.environ.get("FALCON_AMD_ALIGN1"):
            traffic, traffic_src = None, "the PMC measurement on file is of k_align2, this run used k_align"
        dom_name = KERNEL_NAME.get(domk, domk)
        if ((domk == "k_align" and os.environ.get("FALCON_AMD_ALIGN1")) or
                (domk == "k_links" and os.environ.get("FALCON_AMD_LINKS1")) or
                (domk == "k_score" and os.environ.get("FALCON_AMD_SCORE1"))):
            dom_name = domk + " (the kernel behind the default one)"
        res = {
            "metric": "consensus_bases_per_sec",
            "value": round(bases_all * args.steps / elapsed, 1),
            "unit": "bases/s",
            "n_gpus": ranks_ran, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (2-bit packed bases, integer DP)",
            "data": "synthetic",
            "config": {
                "workload": wl["text"],
                "piles_per_step_per_gpu": args.piles,
                "flags": "--min-cov 4 --min-idt 0.70 --max-n-read 200 (K=8)",
                "sequences_per_step_per_gpu": int(st.n_seqs),
                "accepted_alignments_per_step_per_gpu": int(st.n_aligned),
            },
            "pipelined_steps": bool(pipelined),
            "batches_in_flight": (depth if pipelined else 1),
            # every rank on its own clock (the whole-job `value` divides by the slowest rank's time)
            "per_rank": [{"rank": i, "bases_per_sec": round(r[0], 1), "ms_per_step": round(r[1], 3),
                          "k_align_ms": round(r[2], 3)} for i, r in enumerate(per_rank)],
            "piles_per_sec": round(piles_all * args.steps / elapsed, 2),
            "roofline": {
                # (`kernel`: the name a rocprofv3 trace shows; `stage`: the key of kernel_ms / stage_ms)
                "bound": "hbm", "kernel": dom_name, "stage": domk,
                "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5),
                # the peak a plain device copy reaches on this box, and the
                # fraction against that denominator too
                "peak_measured_copy": stream_gbs,
                "frac_of_measured": round(ach / stream_gbs, 5) if stream_gbs else None,
                "algorithmic_bytes_per_launch": int(kalg[domk]),
                "avg_launch_ms": round(kernel_ms[domk], 4),
                # the same kernel launched with nothing beside it (two extra unpipelined
                # passes after the timed region); null when the steps are not pipelined,
                # where avg_launch_ms already is that number
                "alone": ({"avg_launch_ms": round(alone_ms[domk], 4),
                           "achieved": round(kalg[domk] / (alone_ms[domk] * 1e-3) / 1e9, 2),
                           "frac": round(kalg[domk] / (alone_ms[domk] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                          if alone_ms.get(domk) else None),
                "traffic": traffic,
                "traffic_source": traffic_src,
                # what actually binds this kernel: wave-instructions per launch (SQ_INSTS_* passes on
                # file) over this run's launch time, against what the chip issues at most
                "issue": measured_issue(domk, args.piles, args.workload,
                                        alone_ms.get(domk) or kernel_ms[domk]),
            },
            "stage_ms": {s: round(v, 4) for s, v in stage_ms.items()},
            "kernel_ms": {n: round(v, 4) for n, v in kernel_ms.items()},
            # (HIP-event spans on the streams the kernels run on: in pipelined steps they are
            # QUEUE-INCLUSIVE -- a back-stream kernel's span holds its wait for the wave slots
            # the co-running alignment kernel frees, so the spans add up to more than a step;
            # `roofline.alone` and the unpipelined profile under profiles/ hold the kernels' own times)
            "kernel_ms_are": "queue-inclusive event spans" if pipelined else "kernel times (one batch at a time)",
            # the kernels' own times: two unpipelined passes after the timed region (one batch at a time, nothing
            # beside it) -- and what running the steps as a pipeline of `batches_in_flight` batches buys:
            # the sum of those times minus the pipelined step
            "kernel_ms_alone": ({n: round(v, 4) for n, v in alone_ms.items()} if alone_ms else None),
            "pipeline_gain_ms": (round(sum(alone_ms.values()) - elapsed / max(1, args.steps) * 1e3, 3)
                                 if alone_ms else None),
            "host_plan_gap_ms": round(host_gap, 3),
            # what a call of fa_batch_submit costs the calling thread (it queues the front
            # kernels and hands the batch to the context's planner thread): mean and worst
            "host_submit_ms": ({"mean": round(sum(submit_ms) / len(submit_ms), 3), "max": round(max(submit_ms), 3)}
                               if submit_ms else None),
            # the alignment stage about itself (fa_stats): arena bytes, resident wavefronts, and
            # k_align2's iterations with two / one alignment running, band placements,
            # parkings, alignments handed to the general kernel, wide rows
            "align": {"arena_bytes": int(getattr(st, "align_arena_bytes", 0)), "slots": int(getattr(st, "align_slots", 0)),
                      "pair_iterations": int(getattr(st, "align_pair_iterations", 0)),
                      "single_iterations": int(getattr(st, "align_single_iterations", 0)),
                      "placements": int(getattr(st, "align_placements", 0)),
                      "replacements_in_loop": int(getattr(st, "align_replacements", 0)),
                      "parkings": int(getattr(st, "align_parkings", 0)),
                      "handed_back": int(getattr(st, "align_handed_back", 0)),
                      "handed_back_by_cause": {"tape": int(getattr(st, "align_handed_back_tape", 0)),
                                               "wide_rows": int(getattr(st, "align_handed_back_wide", 0)),
                                               "escape_list": int(getattr(st, "align_handed_back_escapes", 0))},
                      "relaunched": int(getattr(st, "align_relaunched", 0)),
                      "wide_rows": int(getattr(st, "align_wide_rows", 0)),
                      "band_rows": int(st.D)},
            # the counts B_alg = L/4 + 4C + 8D + 16A + 12T + 5O is made of, per step and GPU
            "work": {"L": int(st.L), "C": int(st.C), "D": int(st.D), "A": int(st.A), "T": int(st.T),
                     "O": int(st.O)},
            "path_b_alg_bytes_per_step": int(st.b_alg()),
            "path_frac_of_hbm_roofline": round(
                st.b_alg() * world * args.steps / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "setup_s": {"generate": round(t_gen, 2), "stage_to_hbm_incl_pcie": round(t_up, 2),
                        "stage_to_hbm_incl_pcie_again": None},
            # the pile generators every rank forked before it touched the GPU: its share of the container's CPU quota
            "generators": {"processes_per_rank": getattr(args, "gen_procs", None), "ranks": world,
                           "cpu_quota_cores": getattr(args, "cpu_quota", None)},
        }
        gpu_cns = None
        if world == 1 and not (args.no_end_to_end and args.no_cpu_baseline):
            # the consensus strings of the resident batch, for the two comparisons below
            try:
                batch.fetch(False)
                gpu_cns = [batch.result(p) for p in range(batch.n_pile)]
            except Exception:
                gpu_cns = None
        if world == 1:
            # staging again, now that the context's pinned and device staging buffers exist:
            # the steady-state cost of handing a batch of host buffers over (outside `value`)
            try:
                t_up2 = time.perf_counter()
                again = eng.batch(piles)
                t_up2 = time.perf_counter() - t_up2
                again.free()
                res["setup_s"]["stage_to_hbm_incl_pcie_again"] = round(t_up2, 2)
            except Exception:  # informative; never lose the GPU line
                pass
        # the resident batches and the engine go before the end-to-end legs: their workers are
        # processes of their own and find the GPU the way a job of fc_run finds it
        for b in pair:
            b.free()
        pair = []
        eng.close()
        eng = None
        if world == 1 and not args.no_end_to_end:
            try:
                res["end_to_end"] = end_to_end(piles, expect=gpu_cns)
            except Exception as e:  # informative; never lose the GPU line
                res["end_to_end"] = {"piles_per_sec": None, "what": "failed: %r" % (e,)}
            try:  # ... and with a worker that stays on the node (consensus_server)
                res["end_to_end_served"] = end_to_end_served(piles, expect=gpu_cns)
            except Exception as e:
                res["end_to_end_served"] = {"piles_per_sec": None, "what": "failed: %r" % (e,)}
        if world > 1 and not args.no_end_to_end:
            # N streams through the multi-stream worker over the N GPUs (the other ranks are
            # done and idle; their resident batches only hold memory)
            try:
                res["end_to_end"] = end_to_end_multi(piles, world)
            except Exception as e:
                res["end_to_end"] = {"piles_per_sec": None, "what": "failed: %r" % (e,)}
            # ... and the way fc_run feeds the node: N single-stream jobs at once, each on the GPU it
            # finds for itself (per worker: device, wall time, piles/s, text rate)
            try:
                res["end_to_end_workers"] = end_to_end_workers(piles, world)
                # (worker i is not rank i -- ranks hold resident batches, workers are processes of their
                # own -- but there are as many, one per GPU if the lock slots did their job)
                for row, w in zip(res["per_rank"], res["end_to_end_workers"].get("workers", [])):
                    row["e2e_worker_device"] = w["devices"]
                    row["e2e_worker_wall_s"] = w["wall_s"]
                    row["e2e_worker_piles_per_sec"] = w["piles_per_sec"]
                    row["e2e_worker_text_GB_per_sec"] = w["text_GB_per_sec"]
            except Exception as e:
                res["end_to_end_workers"] = {"piles_per_sec": None, "what": "failed: %r" % (e,)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"], cpu_cns = cpu_baseline(
                    piles, timed_per_worker=max(1, args.cpu_baseline_timed),
                    procs=[x for x in args.cpu_baseline_procs.split(",") if x.strip()] or None)
                if gpu_cns is not None:
                    bad = [i for i, s in cpu_cns.items() if gpu_cns[i] != s]
                    res["parity_checked_piles"] = len(cpu_cns)
                    res["parity_mismatches"] = len(bad)
                    res["parity_against"] = res["cpu_baseline"]["kind"]
                    if bad:
                        res["parity_mismatching_piles"] = bad[:16]
            except Exception as e:  # the baseline is informative; never lose the GPU line
                res["cpu_baseline"] = {"value": None, "unit": "bases/s", "cores": 0,
                                       "kind": "unavailable", "sample": "failed: %r" % (e,)}
        print(json.dumps(res), file=out, flush=True)
    for b in pair:
        b.free()
    if eng is not None:
        eng.close()
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.workload == "e2e-long":
        # (SURVEY.md 8d "end-to-end", on a stream long enough that start-up is a few per cent of it: its own line,
        # never the headline's `value`)
        from benchlib.e2e import end_to_end_long
        from benchlib.workloads import gen_piles
        wl = WORKLOADS["ecoli"]
        n = args.piles if args.piles > 0 else wl["piles"]
        piles = gen_piles(range(1000, 1000 + n), max(1, (os.cpu_count() or 2) // 2), wl)
        res = end_to_end_long(piles, jobs=args.jobs, parallel=args.parallel)
        best = max((r for r in res["settings"] if r.get("piles_per_sec")), key=lambda r: r["piles_per_sec"])
        print(json.dumps({"metric": "end-to-end piles/sec, LA4Falcon text -> FASTA through one consensus server, long stream",
                          "value": best["piles_per_sec"], "unit": "piles/s", "n_gpus": 1, "higher_is_better": True,
                          "data": "synthetic", "config": {"workload": wl["text"], "jobs": args.jobs, "parallel": args.parallel},
                          "end_to_end_long": res}))
        return
    if args.workload not in WORKLOADS:
        if args.gpus != 1:
            sys.exit("bench.py --workload %s: one GPU (the 8(f) paths have no multi-GPU story of their own)" % args.workload)
        from benchlib import secondary
        secondary.run(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args, argv)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)"
                 % (args.gpus, world))
    wl = WORKLOADS[args.workload]
    # synthetic input first (forks worker processes; no GPU state exists yet)
    # (worker processes: what this rank's share of the container's CPU quota carries -- 8 ranks forking
    # 32 generators each on a 16-core quota would be 256 runnable processes)
    from benchlib.cpu_baseline import _cgroup_cpu
    ncpu = os.cpu_count() or 1
    quota = _cgroup_cpu()[0]
    if quota:
        ncpu = max(1, min(ncpu, int(quota)))
    procs = max(1, min(32, ncpu // max(1, world)))
    args.gen_procs, args.cpu_quota = procs, (float(quota) if quota else None)
    seeds = [1000003 * (rank + 1) + i for i in range(args.piles)]
    t0 = time.perf_counter()
    piles = gen_piles(seeds, procs, wl)
    gen_s = time.perf_counter() - t0
    plumb = Plumbing(rank, local_rank, world, "nccl")
    from falcon_amd.engine import Engine
    try:
        bench_rank(args, plumb, Engine, piles, gen_s)
    finally:
        plumb.close()


if __name__ == "__main__":
    main()
