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
for N in (1, 2, 4, 8):
    we = W // N
    def run(k):
        infl = []
        for _ in range(k):
            infl.append(G.msm_device_async(srs, d.data_ptr(), n, 0, 0, we))
            if len(infl) == 2:
                G.msm_wait(infl.pop(0))
        while infl:
            G.msm_wait(infl.pop(0))
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); run(20); dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(5): G.msm_wait(G.msm_device_async(srs, d.data_ptr(), n, 0, 0, we))
    lat = (time.perf_counter() - t0) / 5
    print("N=%d: %2d windows per rank: %.3f ms/step pipelined (speed-up %.2fx of N), %.3f ms latency" % (N, we, dt * 1e3, 0, lat * 1e3), flush=True)
