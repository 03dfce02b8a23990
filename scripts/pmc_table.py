#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc passes (counter_collection CSVs under a directory of
passes, one counter set per pass): mean counter value per dispatch of every k_* kernel, and
the ratios that need no unit knowledge (LDS bank-conflict cycles / LDS busy cycles, waiting
/ resident wave cycles, L2 hit rate, instructions per wavefront).

    python scripts/pmc_table.py gpurun_out/pmc8 [gpurun_out/pmc7 ...]
"""
import collections
import csv
import glob
import json
import os
import sys


def bench_line(d):
    for f in sorted(glob.glob(os.path.join(d, "p*.log"))):
        for line in open(f, errors="replace"):
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    pass
    return None


def table(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("k_"):
                agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    j = bench_line(d)
    print("=" * 100)
    print("passes under %s" % d)
    if j:
        print("  bench line of the profiled build: %d piles per launch, stage ms %s%s"
              % (j["config"]["piles_per_step_per_gpu"], j.get("stage_ms"),
                 ", kernel ms %s" % j["kernel_ms"] if "kernel_ms" in j else ""))
    for k in sorted(agg):
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        print("\n%s   (%d dispatch(es) per pass)" % (k, max(len(v) for v in agg[k].values())))
        for n in sorted(c):
            print("   %-30s %.6g" % (n, c[n]))
        r = []
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r.append("LDS bank-conflict cycles / LDS busy cycles = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in c:
            r.append("wave cycles stalled at issue (SQ_WAIT_INST_ANY: pipe busy, dependency) = %.3f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
        if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
            r.append("wave cycles parked at s_waitcnt / a barrier (SQ_WAIT_ANY) = %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            r.append("L2 hit rate = %.3f" % (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and "TCP_TCC_READ_REQ_sum" in c:
            r.append("vector L1 read requests passed on to L2 / L1 accesses = %.3f" % (c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]))
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"):
                if n in c:
                    r.append("%s per wavefront = %.4g" % (n[3:], c[n] / c["SQ_WAVES"]))
        for x in r:
            print("   -> " + x)


for d in sys.argv[1:]:
    table(d)
