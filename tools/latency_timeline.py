#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/latency_run.py: the kernels of ONE MSM from the middle of the run -- start (us, relative to the
MSM's first kernel), duration (us), gap to the previous kernel's end (us), kernel.   usage: latency_timeline.py <dir>"""
import csv, glob, os, re, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
fin = [i for i, r in enumerate(rows) if "msm_final" in r[2]]
a, b = fin[len(fin) // 2 - 1] + 1, fin[len(fin) // 2]
t0, prev = rows[a][0], rows[a][0]
for s, e, name in rows[a:b + 1]:
    short = re.sub(r"\(.*", "", name).replace("bbgpu::", "")
    print("%8.1f  %7.1f  %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short[:70]))
    prev = e
print("first start -> last end: %.1f us; to the next MSM's first kernel: %.1f us" % ((rows[b][1] - t0) / 1e3, (rows[b + 1][0] - t0) / 1e3 if b + 1 < len(rows) else -1))
