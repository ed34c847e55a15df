#!/usr/bin/env python3
"""one-box comparison of N-way splits of a 2^20 MSM: a middle 1/N share as a ROW range (window-major rows of the table: ~15/N windows of all points) against a
POINT range (all 15 windows of n/N points: the digit kernel converts n/N scalars instead of n), 1 .. 8 in flight.  usage: point_share_ab.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
W = G.srs_num_windows(srs, n)
def timed(issue, depth):
    def run(k):
        infl = []
        for _ in range(k):
            infl.append(issue())
            if len(infl) == depth: G.msm_wait(infl.pop(0))
        while infl: G.msm_wait(infl.pop(0))
    run(30); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(24); ts.append((time.perf_counter() - t0) / 24)
    return float(np.median(ts)) * 1e3
full = timed(lambda: G.msm_device_async(srs, d.data_ptr(), n), 2)
print("whole MSM, 2 in flight: %.4f ms/step" % full, flush=True)
for N in [int(a) for a in (sys.argv[1:] or ["8", "4", "2"])]:
    rows = W * n // N
    m = n // N
    off = m * (N // 2)
    for name, issue in (("rows  ", lambda: G.msm_device_rows_async(srs, d.data_ptr(), n, rows * (N // 2), rows * (N // 2 + 1))),
                        ("points", lambda: G.msm_device_async(srs, d.data_ptr() + off * 32, m, off))):
        ts = [(dp, timed(issue, dp)) for dp in (1, 2, 3, 4, 6)]
        print("%s N=%d:" % (name, N) + "".join("  %d in flight %.4f" % t for t in ts) + "   best = %.2fx of the whole" % (full / min(t[1] for t in ts)), flush=True)
# the shares add up: sum of the N point-range results against the one-call result
whole = G.msm_device(srs, d.data_ptr(), n)
N = 8
parts = [G.msm_device(srs, d.data_ptr() + (n // N) * r * 32, n // N, (n // N) * r) for r in range(N)]
print("8 point-range shares add up to the whole:", bool(np.array_equal(G.g1_sum(np.stack(parts)), whole)) if hasattr(G, "g1_sum") else "n/a", flush=True)
