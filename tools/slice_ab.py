#!/usr/bin/env python3
"""a rank's share of a point-range split as its own SRS: n / N points x^(first + i) G with their own window tables (BBGPU_TABLE_C picks the window size),
1 .. 6 in flight; checked against the same point range of the whole table.  usage: slice_ab.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = n // N
off = m * (N // 2)
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
G.set_point_share(N)  # as bench.py --shard points: the window size of the whole MSM (BBGPU_TABLE_C still overrides)
srs = G.srs_generate(x, m, first=off)
G.set_point_share(1)
W = G.srs_num_windows(srs, m)
def timed(issue, depth):
    def run(k):
        infl = []
        for _ in range(k):
            infl.append(issue())
            if len(infl) == depth: G.msm_wait(infl.pop(0))
        while infl: G.msm_wait(infl.pop(0))
    run(30); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(24); ts.append((time.perf_counter() - t0) / 24)
    return float(np.median(ts)) * 1e3
ts = [(dp, timed(lambda: G.msm_device_async(srs, d.data_ptr() + off * 32, m), dp)) for dp in (1, 2, 3, 4, 6)]
print("slice N=%d (%d points, %d windows):" % (N, m, W) + "".join("  %d in flight %.4f" % t for t in ts), flush=True)
mine = G.msm_device(srs, d.data_ptr() + off * 32, m)
G.set_table_share(0, 1)
full = G.srs_generate(x, n)
print("equals the same point range of the whole table:", bool(np.array_equal(mine, G.msm_device(full, d.data_ptr() + off * 32, m, off))), flush=True)
