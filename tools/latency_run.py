#!/usr/bin/env python3
"""one MSM at a time (depth 1), 30 of them, at 2^argv[1] points: the workload for a latency timeline under rocprofv3 --kernel-trace"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
for _ in range(30):
    G.msm_wait(G.msm_device_async(srs, d.data_ptr(), n))
torch.cuda.synchronize()
