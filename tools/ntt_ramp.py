#!/usr/bin/env python3
"""how long after an idle period the resident transform reaches its steady rate: batches of back-to-back ffts, HIP events per batch
    python tools/ntt_ramp.py [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
s = torch.cuda.Stream()
idle = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.5
for lg, per, batches in ((20, 20, 16), (22, 5, 16)):
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
G.shutdown()
