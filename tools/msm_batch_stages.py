#!/usr/bin/env python3
"""stage times of a J-job batched MSM pass (the prover's commitment rounds) at 2^20 points"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
ds = []
for j in range(3):
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    ds.append(torch.from_numpy(sc.view(np.int64)).cuda())
names = ("total", "digits", "sort", "accumulate", "merge", "rowcol", "final")
G.set_timing(True)
for jobs in (1, 2, 3):
    ptrs = [d.data_ptr() for d in ds[:jobs]]
    acc = np.zeros(7); wall = 0.0
    for it in range(6):
        t0 = time.perf_counter(); G.msm_batch_wait(G.msm_device_batch_async(srs, ptrs, n)); dt = time.perf_counter() - t0
        if it >= 2:
            acc += np.array(G.last_timing()[:7]); wall += dt
    print("jobs %d (W=%d): wall %.3f ms | " % (jobs, G.srs_num_windows(srs, n), wall / 4 * 1e3) + "  ".join("%s %.3f" % (k, v / 4) for k, v in zip(names, acc)), flush=True)
