#!/usr/bin/env python3
"""What one rank of an N-way sharded 2^20 MSM costs (no exchange step): rank 0's share on this GPU -- a row range [0, W n / N) as bench.py
issues it, and whole windows [0, ceil(W / N)) as the fallback without tables -- 1 .. 4 in flight.  Upper bound of the N-GPU strong-scaling
curve when the 96-byte all-gather is free: the stated PREDICTION for SCALE_rNN.json (profiles/r02_shard_prediction.json; argv[1] = output path)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu

G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
W = G.srs_num_windows(srs, n)
base = None
report = {"what": "one rank's share of a 2^20-point MSM on one MI355X, ms per step (median of 5 x 24 steps), no exchange step", "windows": W, "point_range": {}, "bucket_range": {}, "row_range": {}, "whole_windows": {}}


def timed(run):
    run(30); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(24); ts.append((time.perf_counter() - t0) / 24)
    return float(np.median(ts))


for N in (1, 2, 4, 8):
    wN = -(-W // N)

    def run_w(k):
        infl = []
        for _ in range(k):
            infl.append(G.msm_device_async(srs, d.data_ptr(), n, 0, 0, wN))
            if len(infl) == 2:
                G.msm_wait(infl.pop(0))
        while infl:
            G.msm_wait(infl.pop(0))
    report["whole_windows"]["N=%d" % N] = {"windows": wN, "ms_per_step_2_in_flight": timed(run_w) * 1e3}
for N in (1, 2, 4, 8):
    rows = W * n // N  # bench.py gives every rank W n / N table rows (a share need not end at a window boundary)
    for depth in (1, 2, 3, 4):
        def run(k):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_rows_async(srs, d.data_ptr(), n, 0, rows))
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        dt = timed(run)
        if base is None and depth == 2:
            base = dt
        report["row_range"].setdefault("N=%d" % N, {})["ms_per_step_%d_in_flight" % depth] = dt * 1e3
        if depth == 2:
            report["row_range"]["N=%d" % N]["speedup_vs_N1"] = base / dt
        print("N=%d (%.3f windows per rank), %d in flight: %.3f ms/step%s" % (N, W / N, depth, dt * 1e3, "  = %.2fx of N=1" % (base / dt) if base else ""), flush=True)

# bucket-range shares (round 3): share 0 of N over all windows and points -- 1 / N of the additions and 1 / N of the buckets to merge and fold
bbase = None
for N in (1, 2, 4, 8):
    for depth in (1, 2, 3, 4):
        def run(k):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_buckets_async(srs, d.data_ptr(), n, N // 2, N))  # a middle share
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        dt = timed(run)
        if bbase is None and depth == 2:
            bbase = dt
        report["bucket_range"].setdefault("N=%d" % N, {})["ms_per_step_%d_in_flight" % depth] = dt * 1e3
        if depth == 2:
            report["bucket_range"]["N=%d" % N]["speedup_vs_N1"] = bbase / dt
        print("buckets N=%d, %d in flight: %.3f ms/step%s" % (N, depth, dt * 1e3, "  = %.2fx of N=1" % (bbase / dt) if bbase else ""), flush=True)

# point-range shares (bench.py's default from round 3 on): a middle rank's n / N points as its own SRS (bbgpu_set_point_share + bbgpu_srs_generate_range), all windows
pbase = None
for N in (1, 2, 4, 8):
    m = n // N
    off = m * (N // 2)
    G.set_point_share(N)
    slice_srs = G.srs_generate(x, m, first=off) if N > 1 else srs
    G.set_point_share(1)
    for depth in (1, 2, 3, 4):
        def run(k):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_async(slice_srs, d.data_ptr() + off * 32, m))
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        dt = timed(run)
        if pbase is None and depth == 2:
            pbase = dt
        report["point_range"].setdefault("N=%d" % N, {})["ms_per_step_%d_in_flight" % depth] = dt * 1e3
        if pbase:
            report["point_range"]["N=%d" % N]["speedup_vs_N1_%d_in_flight" % depth] = pbase / dt
        print("points N=%d, %d in flight: %.3f ms/step%s" % (N, depth, dt * 1e3, "  = %.2fx of N=1" % (pbase / dt) if pbase else ""), flush=True)
    if N > 1:
        G.srs_release(slice_srs)

# host-side cost of one step at the smallest share: time inside the two calls
rows8 = W * n // 8
ti = tw = 0.0
infl = []
for _ in range(40):
    t0 = time.perf_counter(); infl.append(G.msm_device_rows_async(srs, d.data_ptr(), n, 0, rows8)); ti += time.perf_counter() - t0
    if len(infl) == 2:
        t0 = time.perf_counter(); G.msm_wait(infl.pop(0)); tw += time.perf_counter() - t0
while infl:
    G.msm_wait(infl.pop(0))
print("N=8 share, 2 in flight: %.3f ms inside msm_device_async (launches), %.3f ms inside msm_wait (event wait + host finish) per step" % (ti / 40 * 1e3, tw / 40 * 1e3))
report["host_side_N8"] = {"ms_in_issue": ti / 40 * 1e3, "ms_in_wait": tw / 40 * 1e3}
for k, v in report["whole_windows"].items():
    print("whole windows %s: %d windows, %.3f ms/step" % (k, v["windows"], v["ms_per_step_2_in_flight"]))
if len(sys.argv) > 1:
    json.dump(report, open(sys.argv[1], "w"), indent=1)
