#!/usr/bin/env python3
"""Host time of bbgpu_msm_g1_batch_wait AFTER the device work has finished (the batch has been given 3 ms): the event query, the Horner
fold of the 15 bit-sliced sums per job and the one normalisation per batch -- what stands between the last kernel of a commitment
round and the prover's next challenge."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
n = 1 << 16
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
ds = []
for j in range(3):
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    ds.append(torch.from_numpy(sc.view(np.int64)).cuda())
for jobs in (1, 2, 3):
    ts = []
    for _ in range(30):
        t = G.msm_device_batch_async(srs, [d.data_ptr() for d in ds[:jobs]], n)
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 0.003:  # busy: a sleeping core wakes up cold
            pass
        t0 = time.perf_counter(); G.msm_batch_wait(t); ts.append(time.perf_counter() - t0)
    print("%d job(s): %.1f us inside batch_wait on a finished batch (median of 30, incl. ~3 us of ctypes)" % (jobs, float(np.median(ts)) * 1e6), flush=True)
