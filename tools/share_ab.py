#!/usr/bin/env python3
"""one-box comparison of the two N-way splits of a 2^20 MSM at N = 8 (and N = 4): a middle share, 1 .. 4 in flight; env knobs apply (BBGPU_ACC_WGS ...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
W = G.srs_num_windows(srs, n)
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
for N in [int(a) for a in (sys.argv[1:] or ["8"])]:
    rows = W * n // N
    r = "rows    N=%d:" % N + "".join("  %d in flight %.3f" % (dp, timed(lambda: G.msm_device_rows_async(srs, d.data_ptr(), n, rows * (N // 2), rows * (N // 2 + 1)), dp)) for dp in (1, 2, 3, 4, 6, 8))
    b = "buckets N=%d:" % N + "".join("  %d in flight %.3f" % (dp, timed(lambda: G.msm_device_buckets_async(srs, d.data_ptr(), n, N // 2, N), dp)) for dp in (1, 2, 3, 4, 6, 8))
    print(r, flush=True); print(b, flush=True)
