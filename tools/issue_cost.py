#!/usr/bin/env python3
"""host time of ONE issue call (all launches of an MSM enqueued, nothing waited for) and of one wait on a finished MSM: what a rank of an N-way split
pays per step on its one host thread, whatever the GPU does meanwhile"""
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
for N in (1, 8):
    rows = W * n // N
    issue = (lambda: G.msm_device_rows_async(srs, d.data_ptr(), n, rows * (N // 2), rows * (N // 2 + 1)))
    for _ in range(3):
        G.msm_wait(issue())
    ti, tw = [], []
    for rep in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tk = [issue() for _ in range(6)]
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        time.sleep(0.01)
        t2 = time.perf_counter()
        for k in tk: G.msm_wait(k)
        t3 = time.perf_counter()
        ti.append((t1 - t0) / 6); tw.append((t3 - t2) / 6)
    print("N=%d share: issue %.1f us per MSM (median of 10 x 6 back-to-back issues), wait on a finished one %.1f us" % (N, np.median(ti) * 1e6, np.median(tw) * 1e6), flush=True)
