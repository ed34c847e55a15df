#!/usr/bin/env python3
"""Why does bench.py's pipelined step differ from tools/msm_ab.py's on the same box?  Same loop, varying one thing at a time:
scalar distribution (uniform residues vs SURVEY 8d's < 2^252 integers in Montgomery form), timing level (0 / 2 / 1), run-to-run spread."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from barretenberg_amd import BbGpu

G = BbGpu(0)
n = 1 << 20
dev = torch.device("cuda", 0)
x = bench.limbs_of(12345678901234567890123456789 % bench.FR_MODULUS)
srs = G.srs_generate(x, n)
rng = np.random.default_rng(7)
uni = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); uni[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d_uni = torch.from_numpy(uni.view(np.int64)).cuda()
d_sm = bench.to_montgomery_on_device(G, bench.raw_scalars(n, bench.SPLITMIX_GAMMA), dev)


def run(d, k, depth=2):
    infl = []
    for _ in range(k):
        infl.append(G.msm_device_async(srs, d.data_ptr(), n))
        if len(infl) == depth:
            G.msm_wait(infl.pop(0)); G.last_timing()
    while infl:
        G.msm_wait(infl.pop(0)); G.last_timing()


for name, d in (("uniform residues", d_uni), ("splitmix < 2^252, Montgomery", d_sm)):
    for level in (0, 2, 1):
        G.set_timing(level)
        run(d, 4); torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(d, 20); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print("%-30s timing level %d: ms/step %s  (min %.4f median %.4f)" % (name, level, " ".join("%.4f" % t for t in ts), min(ts), float(np.median(ts))), flush=True)
G.set_timing(0)
