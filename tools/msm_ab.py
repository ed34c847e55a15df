#!/usr/bin/env python3
"""one-box A/B helper (medians of 5 x 20 steps): single-MSM latency, two-in-flight step time at the full and at the 1/8 window share, 2^16 latency, prover 2^16"""
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
            run(40 if lg == 20 and we == W else 20, depth); torch.cuda.synchronize()  # the first ~50 ms after idle run ~4 % slow (clock ramp)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); run(20, depth); ts.append((time.perf_counter() - t0) / 20)
            out.append(float(np.median(ts)) * 1e3)
        print("2^%d, %2d of %d windows: latency %.4f ms, two in flight %.4f ms/step, three in flight %.4f ms/step" % (lg, we, W, out[0], out[1], out[2]), flush=True)
