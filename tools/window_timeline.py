#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: every kernel that starts inside a window of `span_us` microseconds beginning at fraction `frac` of the
trace -- start (us, relative), duration (us), gap to the previous end (us), queue, kernel.   usage: window_timeline.py <dir> [frac] [span_us]"""
import csv, glob, os, re, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
span = float(sys.argv[3]) if len(sys.argv) > 3 else 3000.0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
t0 = rows[0][0] + int((rows[-1][1] - rows[0][0]) * frac)
prev = None
for s, e, q, name in rows:
    if t0 <= s <= t0 + span * 1e3:
        short = re.sub(r"\(.*", "", name).replace("bbgpu::", "").replace("void ", "")
        print("%8.1f  %7.1f  %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, q, short[:64]))
        prev = max(prev or e, e)
