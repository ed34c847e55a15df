#!/usr/bin/env python3
"""workload for a timeline under rocprofv3 --kernel-trace: a middle 1/N share of a 2^20 MSM, `depth` in flight, 40 steps.
usage: share_run.py rows|buckets|points N depth"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
kind, N, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n) if kind != "points" else None
if kind == "points":  # as bench.py --shard points does it: the middle rank's slice as its own SRS
    G.set_point_share(N)
    slice_srs = G.srs_generate(x, n // N, first=(n // N) * (N // 2))
    G.set_point_share(1)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
W = G.srs_num_windows(srs, n) if srs is not None else 0
rows = W * n // N
infl = []
for _ in range(40):
    if kind == "rows": infl.append(G.msm_device_rows_async(srs, d.data_ptr(), n, rows * (N // 2), rows * (N // 2 + 1)))
    elif kind == "points": infl.append(G.msm_device_async(slice_srs, d.data_ptr() + (n // N) * (N // 2) * 32, n // N))
    else: infl.append(G.msm_device_buckets_async(srs, d.data_ptr(), n, N // 2, N))
    if len(infl) == depth: G.msm_wait(infl.pop(0))
while infl: G.msm_wait(infl.pop(0))
torch.cuda.synchronize()
