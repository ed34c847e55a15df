#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (the drop-in boundary hands over pageable host buffers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(1)
x = rng.integers(0, 1 << 64, size=(1, 4), dtype=np.uint64); x[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
h, table = G.srs_generate(x[0], n, want_host_table=True)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
G.pippenger(sc, table, n)
t0 = time.perf_counter()
for _ in range(10): G.pippenger(sc, table, n)
dt = (time.perf_counter() - t0) / 10
print("bbgpu_msm_g1 (host scalars, resident SRS) 2^20: %.3f ms  %.3e points/s" % (dt * 1e3, n / dt))
jobs = [(table, sc, n)] * 3
G.batched_scalar_multiplications(jobs)
t0 = time.perf_counter()
for _ in range(5): G.batched_scalar_multiplications(jobs)
dt = (time.perf_counter() - t0) / 5 / 3
print("bbgpu_msm_g1_batch (3 jobs, pipelined) per MSM: %.3f ms  %.3e points/s" % (dt * 1e3, n / dt))
co = sc.copy()
G.fft(co)
t0 = time.perf_counter()
for _ in range(10): G.fft(co)
dt = (time.perf_counter() - t0) / 10
print("bbgpu_ntt (host buffer, H2D + D2H) 2^20: %.3f ms  %.3e elements/s" % (dt * 1e3, n / dt))
