#!/usr/bin/env python3
"""stage times of one 2^20-point MSM for the skewed scalar sets of bench.py (all equal / {0, 1, -1} / values below 200) next to uniform ones"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
x = np.array([5, 6, 7, 8], dtype=np.uint64)
srs = G.srs_generate(x, n)
rng = np.random.default_rng(7)
uni = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
sets = {"uniform": uni}
for k in bench.SKEWED_KINDS:
    sets[k] = np.ascontiguousarray(bench.skewed_scalars(k, n))
for name, sc in sets.items():
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    for _ in range(5): G.msm_device(srs, d.data_ptr(), n)
    G.set_timing(True)
    acc = np.zeros(7)
    for _ in range(5):
        G.msm_device(srs, d.data_ptr(), n); acc += np.array(G.last_timing()[:7])
    G.set_timing(False)
    print("%-10s total %.3f digits %.3f sort %.3f acc %.3f merge %.3f folds %.3f collect %.3f" % ((name,) + tuple(acc / 5)), flush=True)
