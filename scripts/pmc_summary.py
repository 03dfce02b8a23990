#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
