#!/usr/bin/env python3
"""one-box A/B helper for the accumulation kernel alone: mean stage time of msm_accumulate_kernel over 20 one-at-a-time 2^20-point MSMs
(HIP events on the launch stream, bbgpu_set_timing(1)), plus the two-in-flight step.  BBGPU_LIB selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
for _ in range(60):
    G.msm_device(srs, d.data_ptr(), n)
G.set_timing(True)
acc = []
for _ in range(20):
    G.msm_device(srs, d.data_ptr(), n)
    acc.append(G.last_timing()[3])
G.set_timing(False)
def run(k, depth=2):
    infl = []
    for _ in range(k):
        infl.append(G.msm_device_async(srs, d.data_ptr(), n))
        if len(infl) == depth:
            G.msm_wait(infl.pop(0))
    while infl:
        G.msm_wait(infl.pop(0))
run(40)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); run(20); ts.append((time.perf_counter() - t0) / 20)
print("%s: accumulate alone %.4f ms (min %.4f), two in flight %.4f ms/step" % (os.path.basename(os.environ.get("BBGPU_LIB", "libbbgpu.so")), float(np.mean(acc)), float(np.min(acc)), float(np.median(ts)) * 1e3), flush=True)
