#!/usr/bin/env python3
"""how long after an idle period the resident kernels reach their steady rate: batches of back-to-back ffts (HIP events per batch), then batches of
two-in-flight 2^20-point MSM steps (wall clock per batch)
    python tools/ntt_ramp.py [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from barretenberg_amd import BbGpu
G = BbGpu(0)
s = torch.cuda.Stream()
idle = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.5
for lg, per, batches in ((20, 50, 30), (22, 12, 30)):
    n = 1 << lg
    x = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    for kind in ("fft", "ifft"):
        G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)  # tables
    torch.cuda.synchronize()
    for kind in ("fft", "ifft", "fft"):
        time.sleep(idle)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(batches + 1)]
        ev[0].record(s)
        for b in range(batches):
            for _ in range(per): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
            ev[b + 1].record(s)
        torch.cuda.synchronize()
        ms = [ev[b].elapsed_time(ev[b + 1]) / per for b in range(batches)]
        t = np.cumsum([m * per for m in ms])
        print("2^%d %-4s after %.0f ms idle: ms per transform by batch of %d: %s   (elapsed at batch ends, ms: %s)" %
              (lg, kind, idle * 1e3, per, " ".join("%.4f" % m for m in ms), " ".join("%.0f" % v for v in t)), flush=True)
    del d
n = 1 << 20
srs = G.srs_generate(bench.limbs_of(12345678901234567890123456789 % bench.FR_MODULUS), n)
d_sm = bench.to_montgomery_on_device(G, bench.raw_scalars(n, bench.SPLITMIX_GAMMA), torch.device("cuda", 0))
for rep in range(3):
    time.sleep(idle)
    infl, ts, t0 = [], [], time.perf_counter()
    for k in range(200):
        infl.append(G.msm_device_async(srs, d_sm.data_ptr(), n))
        if len(infl) == 2:
            G.msm_wait(infl.pop(0))
        if k % 10 == 9:
            t1 = time.perf_counter(); ts.append((t1 - t0) / 10 * 1e3); t0 = t1
    while infl:
        G.msm_wait(infl.pop(0))
    print("2^20-point MSM, two in flight, after %.0f ms idle: ms per step by batch of 10 steps: %s" % (idle * 1e3, " ".join("%.4f" % v for v in ts)), flush=True)
G.shutdown()
