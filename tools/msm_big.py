#!/usr/bin/env python3
"""MSM beyond one window-table segment (2^21 .. 2^24 points: one table per <= 2^20-point segment, point-range pieces on two slots, piece sums
added on the host): additivity check, wall time of one call, two calls in flight, one-time table cost.  BBGPU_PRECOMPUTE=0: the per-window
bucket sets of rounds 1-3 beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
if os.environ.get("BBGPU_PRECOMPUTE") == "0":
    G.set_precompute(False)
rng = np.random.default_rng(3)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
lgs = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24]
for lg in lgs:
    n = 1 << lg
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter(); h = G.srs_generate(x, n); t_srs = time.perf_counter() - t0
    held = free0 - torch.cuda.mem_get_info()[0]
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    full = G.msm_device(h, d.data_ptr(), n)
    for _ in range(2): G.msm_device(h, d.data_ptr(), n)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); full2 = G.msm_device(h, d.data_ptr(), n); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    # two calls in flight (what a prover's consecutive commitments look like)
    reps = 6
    for _ in range(2):  # the slots the second ticket and its helper land on allocate their workspaces on first use: not in the timed loop
        tw = [G.msm_device_async(h, d.data_ptr(), n), G.msm_device_async(h, d.data_ptr(), n)]
        for t in tw: G.msm_wait(t)
    t0 = time.perf_counter()
    tk = [G.msm_device_async(h, d.data_ptr(), n)]
    for _ in range(reps - 1):
        tk.append(G.msm_device_async(h, d.data_ptr(), n))
        G.msm_wait(tk.pop(0))
    G.msm_wait(tk.pop(0))
    dt2 = (time.perf_counter() - t0) / reps
    m = n // 2 + 777
    lo = G.msm_device(h, d.data_ptr(), m)
    hi = G.msm_device(h, d.data_ptr() + m * 32, n - m, offset=m)
    ok = np.array_equal(G.g1_sum(np.stack([lo, hi])), full) and np.array_equal(full, full2)
    print("2^%d (%d windows, tables %s): additivity %s, one call %.2f ms (min %.2f) = %.3e points/s, two in flight %.2f ms per call = %.3e points/s | SRS + tables once: %.0f ms, %.2f GiB"
          % (lg, G.srs_num_windows(h, n), G.srs_has_window_tables(h), ok, dt * 1e3, min(ts) * 1e3, n / dt, dt2 * 1e3, n / dt2, t_srs * 1e3, held / 2**30), flush=True)
    G.srs_release(h); del d
