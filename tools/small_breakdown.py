#!/usr/bin/env python3
"""where a small host-pointer MSM (table not resident: the verifier's case) spends its wall time: device stages (HIP events) against the call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu
from oracle.pyoracle import Oracle, aligned_copy  # inputs only
O = Oracle()
G = BbGpu(0)
G.set_host_thresholds(0, 0)
srs = O.make_srs(O.random_scalars(7, 1)[0], 1024)
table = O.point_table(srs)
sc = O.random_scalars(9, 1024)
for n in (32, 256, 1000):
    s, t = aligned_copy(sc[:n]), aligned_copy(table[:2 * n])
    for level in (0, 1):
        G.set_timing(level)
        for _ in range(3): G.pippenger(s, t, n)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); G.pippenger(s, t, n); ts.append(time.perf_counter() - t0)
        line = "n=%4d timing level %d: call %.3f ms (median of 15)" % (n, level, sorted(ts)[7] * 1e3)
        if level:
            st = G.last_timing()
            line += "  device stages ms: total %.3f digits %.3f sort %.3f accumulate %.3f merge %.3f folds %.3f final %.3f" % tuple(st[:7])
        print(line, flush=True)
G.set_timing(0)
G.shutdown()
