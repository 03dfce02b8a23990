#!/usr/bin/env python3
"""profiles/pmc_issue.json from rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0` that hold
SQ_INSTS_VALU / _SALU / _LDS / _VMEM_RD / _VMEM_WR of the path's kernels (scripts/pmc_run.sh <tag> 3072 sq):
wave-instructions per launch of the alignment stage's kernel, by class.  bench.py presents them as
`roofline.issue` -- over the launch time of its own run, against the chip's measured issue ceiling
(profiles/r02_ubench_issue_rates.txt) -- while the kernel's source is unchanged (the record carries its digest).

    python scripts/pmc_issue_record.py gpurun_out/<dir with p*/ passes> [stage] [workload]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from benchlib.traffic import KERNEL_NAME
    d = sys.argv[1]
    stage = sys.argv[2] if len(sys.argv) > 2 else "k_align"
    workload = sys.argv[3] if len(sys.argv) > 3 else "ecoli"
    kname = KERNEL_NAME.get(stage, stage)
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if name == kname:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    line = None
    for f in sorted(glob.glob(os.path.join(d, "*.log"))):
        for ln in open(f, errors="replace"):
            if ln.startswith("{"):
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
    if line is None:
        sys.exit("no bench line in the pass logs under " + d)
    mean = lambda c: (sum(agg[c]) / len(agg[c])) if agg.get(c) else None
    need = ("SQ_INSTS_VALU", "SQ_INSTS_SALU")
    if any(mean(c) is None for c in need):
        sys.exit("passes with %s of %s are needed" % (" and ".join(need), kname))
    rec = {
        "kernel": kname, "piles_per_launch": line["config"]["piles_per_step_per_gpu"], "workload": workload,
        "valu": mean("SQ_INSTS_VALU"), "salu": mean("SQ_INSTS_SALU"), "lds": mean("SQ_INSTS_LDS"),  # (None: that pass is not among these)
        "vmem": (mean("SQ_INSTS_VMEM_RD") or 0) + (mean("SQ_INSTS_VMEM_WR") or 0),
        "branch": mean("SQ_INSTS_BRANCH"),
        "wave_cycles": mean("SQ_WAVE_CYCLES"), "wait_any": mean("SQ_WAIT_ANY"), "wait_inst_any": mean("SQ_WAIT_INST_ANY"),
        "source_sha": bench.kernel_source_sha(stage),
        "taken_on": os.path.basename(os.path.normpath(d)),
        "what": "SQ_INSTS_* of %s, means per launch over the dispatches of separate --pmc passes of `bench.py --steps 1 "
                "--warmup 0 --piles %d` (kernel-trace only)" % (kname, line["config"]["piles_per_step_per_gpu"]),
    }
    path = os.path.join(ROOT, "profiles", "pmc_issue.json")
    try:
        allrec = json.load(open(path))
    except (OSError, ValueError):
        allrec = {}
    allrec[stage] = rec
    json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
