#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0`
(one counter set per pass, kernel-trace only): HBM bytes per launch of the dominant kernel
as /opt/skills/guides/MI355X_MICROARCH.md prescribes -- WRITE_SIZE and FETCH_SIZE in separate
passes, both in KiB, each calibrated on a kernel of this very run whose traffic is known
exactly: k_pack streams the staged ASCII in (16 bytes per lane) and the 2-bit words out.

    python scripts/pmc_traffic_record.py gpurun_out/<dir with p*/ passes> [kernel] [workload] [read factor]

FETCH_SIZE counts 64 bytes per request to the fabric: k_pack's wide streaming reads go out
as 128-byte requests and need x1.90 (the guide's "double it"); k_align's reads are the
trace-back's 4-byte gathers, one 64-byte request each (sum of edit distances = 0.58 G
gathers x 64 B = 37.1 GB predicted, 37.7 GB counted), so for k_align the read side is
taken raw (read factor 1.0) -- the k_pack factor is recorded beside it.

The record carries the digest of the kernel's source file; bench.py presents it as
`roofline.traffic` only while that source is unchanged."""
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
    d = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_align"
    workload = sys.argv[3] if len(sys.argv) > 3 else "ecoli"
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if name == "k_align2":   # (the alignment stage of the path, whichever kernel runs it)
                name = "k_align"
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    line = None
    for f in sorted(glob.glob(os.path.join(d, "*.log"))):
        for ln in open(f, errors="replace"):
            if ln.startswith("{"):
                try:
                    line = json.loads(ln)
                except ValueError:
                    pass
    if line is None or "work" not in line:
        sys.exit("no bench line with its work counts in the pass logs under " + d)
    piles = line["config"]["piles_per_step_per_gpu"]
    n_seq = line["config"]["sequences_per_step_per_gpu"]
    L = line["work"]["L"]
    mean = lambda k, c: (sum(agg[k][c]) / len(agg[k][c])) if agg[k].get(c) else None
    # k_pack, known exactly up to the per-sequence padding (engine.hip: 16-byte aligned ASCII
    # + 16; packed words: ceil(len/16) + 2 rounded up to 4 words)
    pack_in = L + n_seq * 24.0
    pack_out = L / 4.0 + n_seq * 14.0
    cal = {}
    for ctr, known in (("FETCH_SIZE", pack_in), ("WRITE_SIZE", pack_out)):
        v = mean("k_pack", ctr)
        cal[ctr] = (known / (v * 1024.0)) if v else None
    rd, wr = mean(kernel, "FETCH_SIZE"), mean(kernel, "WRITE_SIZE")
    if rd is None or wr is None:
        sys.exit("passes for FETCH_SIZE and WRITE_SIZE of %s are both needed" % kernel)
    rd_factor = float(sys.argv[4]) if len(sys.argv) > 4 else (cal["FETCH_SIZE"] or 1.0)
    rd_b = rd * 1024.0 * rd_factor
    wr_b = wr * 1024.0 * (cal["WRITE_SIZE"] or 1.0)
    alg = line["roofline"]["algorithmic_bytes_per_launch"] if line["roofline"]["kernel"] == kernel else None
    rec = {
        "piles_per_launch": piles, "workload": workload,
        "hbm_bytes_per_launch": int(rd_b + wr_b),
        "read_bytes": int(rd_b), "write_bytes": int(wr_b),
        "raw_KiB": {"FETCH_SIZE": rd, "WRITE_SIZE": wr},
        "calibration_on_k_pack": {k: (round(v, 4) if v else None) for k, v in cal.items()},
        "read_factor_applied": rd_factor,
        "algorithmic_bytes_per_launch": alg,
        "source_sha": bench.kernel_source_sha(kernel),
        "taken_on": os.path.basename(os.path.normpath(d)),
        "what": "FETCH_SIZE + WRITE_SIZE (KiB) of %s, separate --pmc passes of `bench.py --steps 1 "
                "--warmup 0` (kernel-trace only, --kernel-include-regex 'k_align|k_pack'); WRITE_SIZE "
                "scaled by the factor that makes k_pack's counter equal its exactly known output "
                "(0.999), FETCH_SIZE by %.2f (k_pack's 16-byte streaming reads need 1.90: 128-byte "
                "requests counted as 64; 4-byte gathers are one 64-byte request each)" % (kernel, rd_factor),
    }
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    # (the E. coli-like workload under the kernel's name, as ever; the others under "<kernel>:<workload>")
    doc[kernel if workload == "ecoli" else "%s:%s" % (kernel, workload)] = rec
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
