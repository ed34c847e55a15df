#!/usr/bin/env python3
"""MSM beyond the window-table limit (2^21 .. 2^24 points, one bucket set per window): additivity check, time and stage split"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
rng = np.random.default_rng(3)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
names = ("total", "digits", "sort", "accumulate", "merge", "rowcol", "final")
for lg in (20, 21, 22, 23, 24):
    n = 1 << lg
    h = G.srs_generate(x, n)
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    full = G.msm_device(h, d.data_ptr(), n)
    G.set_timing(True)
    t0 = time.perf_counter(); full2 = G.msm_device(h, d.data_ptr(), n); dt = time.perf_counter() - t0
    tm = G.last_timing()
    G.set_timing(False)
    m = n // 2 + 777
    lo = G.msm_device(h, d.data_ptr(), m)
    hi = G.msm_device(h, d.data_ptr() + m * 32, n - m, offset=m)
    ok = np.array_equal(G.g1_sum(np.stack([lo, hi])), full) and np.array_equal(full, full2)
    print("2^%d (%d windows): additivity %s, %.2f ms = %.3e points/s | " % (lg, G.srs_num_windows(h, n), ok, dt * 1e3, n / dt)
          + "  ".join("%s %.2f" % (k, v) for k, v in zip(names, tm)), flush=True)
    G.srs_release(h); del d
