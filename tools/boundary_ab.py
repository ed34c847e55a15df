#!/usr/bin/env python3
"""one-box check of the host-pointer entries: bbgpu_msm_g1 at 2^20 (BBGPU_HOST_MSM_SPLIT ranges) and bbgpu_ntt at 2^20, median of 10 after 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu

G = BbGpu(0)
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
srs, table = G.srs_generate(x, n, True)
hs = []
for k in range(3):
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    hs.append(sc)
def med(f, reps=10, warm=3):
    for _ in range(warm): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
turn = [0]
def one():
    turn[0] += 1
    return G.pippenger(hs[turn[0] % 3], table, n)
r0 = one()
print("split=%s  bbgpu_msm_g1 2^%d: %.3f ms" % (os.environ.get("BBGPU_HOST_MSM_SPLIT", "default"), lg, med(one)), flush=True)
co = hs[0].copy()
for kind in ("fft", "coset_fft"):
    print("bbgpu_ntt %s 2^%d: %.3f ms" % (kind, lg, med(lambda: G.ntt(co, kind))), flush=True)
