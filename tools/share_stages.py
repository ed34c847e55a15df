#!/usr/bin/env python3
"""stage times (HIP events on the launch stream) of one rank's window share of a 2^20 MSM, one MSM at a time"""
import os, sys
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
G.set_timing(True)
names = ("total", "digits", "sort", "accumulate", "merge", "rowcol", "final")
for N in (1, 2, 4, 8):
    we = W // N
    acc = np.zeros(7)
    for it in range(8):
        G.msm_wait(G.msm_device_async(srs, d.data_ptr(), n, 0, 0, we))
        if it >= 3:
            acc += np.array(G.last_timing()[:7])
    print("N=%d (%2d windows): " % (N, we) + "  ".join("%s %.3f" % (k, v / 5) for k, v in zip(names, acc)), flush=True)
