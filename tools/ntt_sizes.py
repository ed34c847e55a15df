#!/usr/bin/env python3
"""device-resident NTT timing across sizes (HIP events on the launch stream), steady state: ~80 ms of untimed transforms before every timed batch of 20"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
s = torch.cuda.Stream()
for lg in (12, 14, 16, 18, 20, 21, 22, 23, 24, 26):
    n = 1 << lg
    x = np.random.default_rng(lg).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    for kind in ("fft", "coset_fft"):
        for _ in range(3): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        torch.cuda.synchronize()
        # out of the post-idle ramp first (tools/ntt_ramp.py: the first ~30 ms of back-to-back transforms run up to 15 % slow): ~80 ms of untimed transforms
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record(s)
        for _ in range(4): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        r1.record(s); torch.cuda.synchronize()
        for _ in range(int(80.0 / max(1e-3, r0.elapsed_time(r1) / 4))): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("2^%d %-10s %.4f ms  %.3e elements/s" % (lg, kind, ms, n / (ms * 1e-3)))
