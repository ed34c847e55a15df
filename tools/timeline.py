#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (bench.py steady state): chronological list of the kernels between two accumulation starts
in the middle of the trace -- start (us, relative), duration (us), queue, kernel.   usage: timeline.py <dir> [steps]"""
import csv, glob, os, re, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
acc = [i for i, r in enumerate(rows) if "msm_accumulate_kernel" in r[3]]
mid = len(acc) // 2
lo, hi = acc[mid], acc[min(mid + steps, len(acc) - 1)]
t0 = rows[lo][0]
# include kernels that started a bit before (the other queue's tail)
for s, e, q, name in rows:
    if rows[lo][0] - 400000 <= s <= rows[hi][0]:
        short = re.sub(r"\(.*", "", name).replace("bbgpu::", "")
        print("%9.1f  %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:60]))
