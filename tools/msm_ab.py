#!/usr/bin/env python3
"""one-box A/B helper: single-MSM latency, two-in-flight step time at the full and at the 1/8 window share, 2^16 latency, prover 2^16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
for lg in (20, 16):
    n = 1 << lg
    srs = G.srs_generate(x, n)
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    W = G.srs_num_windows(srs, n)
    for we in (W, max(1, W // 8)):
        def run(k, depth):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_async(srs, d.data_ptr(), n, 0, 0, we))
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        out = []
        for depth in (1, 2, 3):
            run(4, depth); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); run(20, depth); best = min(best, (time.perf_counter() - t0) / 20)
            out.append(best * 1e3)
        print("2^%d, %2d of %d windows: latency %.4f ms, two in flight %.4f ms/step, three in flight %.4f ms/step" % (lg, we, W, out[0], out[1], out[2]), flush=True)
