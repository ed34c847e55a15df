#!/usr/bin/env python3
"""What one rank of an N-way window-sharded 2^20 MSM costs (no exchange step): rank 0's window share on this GPU, two in flight,
as bench.py issues it.  Upper bound of the N-GPU strong-scaling curve when the 96-byte all-gather is free."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
W = G.srs_num_windows(srs, n)
base = None
for N in (1, 2, 4, 8):
    rows = W * n // N  # bench.py gives every rank W n / N table rows (a share need not end at a window boundary)
    for depth in (1, 2, 3, 4):
        def run(k):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_rows_async(srs, d.data_ptr(), n, 0, rows))
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        run(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(24); dt = (time.perf_counter() - t0) / 24
        if base is None and depth == 2:
            base = dt
        print("N=%d (%.3f windows per rank), %d in flight: %.3f ms/step%s" % (N, W / N, depth, dt * 1e3, "  = %.2fx of N=1" % (base / dt) if base else ""), flush=True)

# host-side cost of one step at the smallest share: time inside the two calls
rows8 = W * n // 8
ti = tw = 0.0
infl = []
for _ in range(40):
    t0 = time.perf_counter(); infl.append(G.msm_device_rows_async(srs, d.data_ptr(), n, 0, rows8)); ti += time.perf_counter() - t0
    if len(infl) == 2:
        t0 = time.perf_counter(); G.msm_wait(infl.pop(0)); tw += time.perf_counter() - t0
while infl:
    G.msm_wait(infl.pop(0))
print("N=8 share, 2 in flight: %.3f ms inside msm_device_async (launches), %.3f ms inside msm_wait (event wait + host finish) per step" % (ti / 40 * 1e3, tw / 40 * 1e3))
