#!/usr/bin/env python3
"""Timeline of bench steps in a rocprofv3 kernel trace (csv): dispatches in start order with
queue, duration and the gap since the previous dispatch on the same queue ended.

    python scripts/timeline.py kt_kernel_trace.csv [kt_memory_copy_trace.csv] [first] [last]

`first` / `last` count k_seed_index launches from the end (default 5 and 2: three pipelined
steps of a default bench.py run, whose last two launches are the unpipelined passes of
roofline.alone)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28],
       r.get("Queue_Id", "?")) for r in rows]
if len(sys.argv) > 2 and sys.argv[2] and sys.argv[2] != "-":
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                   "copy:" + r.get("Direction", "?").replace("MEMORY_COPY_", "")[:22], "copy"))
first = int(sys.argv[3]) if len(sys.argv) > 3 else 5
last = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_seed_index")]
lo = starts[-first] if len(starts) >= first else starts[0]
hi = starts[-last] if 0 < last <= len(starts) else len(ev)
t0 = ev[lo][0]
last_end = {}
print("%10s %9s %8s  %-28s %s" % ("start_ms", "dur_ms", "gap_ms", "what", "queue"))
for s, e, name, q in ev[lo:hi]:
    gap = (s - last_end[q]) / 1e6 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    if (e - s) / 1e6 < 0.02 and gap < 0.2:
        continue
    print("%10.3f %9.3f %8.3f  %-28s %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap, name, q))
