#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: the accumulate kernel's per-dispatch duration (which includes the time a dispatch sits in
its queue behind the previous accumulation when two MSMs are in flight) next to the spacing of consecutive END timestamps -- the figure
bench.py reports as roofline.kernel_ms.   usage: acc_spacing.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
acc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if "msm_accumulate_kernel" in r["Kernel_Name"])
dur = [(e - s) / 1e6 for s, e in acc]
gap = [(acc[i][1] - acc[i - 1][1]) / 1e6 for i in range(1, len(acc))]
over = [i for i in range(1, len(acc)) if acc[i][0] < acc[i - 1][1]]  # dispatched while the previous one was still running
print("%d dispatches, mean duration %.3f ms" % (len(acc), sum(dur) / len(dur)))
if over:
    print("%d dispatched while the previous accumulation was running: mean duration %.3f ms, mean end-to-end spacing %.3f ms"
          % (len(over), sum(dur[i] for i in over) / len(over), sum(gap[i - 1] for i in over) / len(over)))
rest = [i for i in range(len(acc)) if i not in over]
print("%d others: mean duration %.3f ms" % (len(rest), sum(dur[i] for i in rest) / len(rest)))
