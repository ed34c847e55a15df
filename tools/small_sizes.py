#!/usr/bin/env python3
"""SURVEY 8b small sizes: wall-clock of the host answer (csrc/host_small.hpp) against the GPU path for the same host-pointer call,
to place BBGPU_HOST_MSM_MAX / BBGPU_HOST_NTT_MAX.  Tables are NOT resident (the verifier builds its points per proof)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu
from oracle.pyoracle import Oracle, aligned_copy  # inputs only

O = Oracle()
G = BbGpu(0)
srs = O.make_srs(O.random_scalars(7, 1)[0], 256)
table = O.point_table(srs)
sc = O.random_scalars(9, 256)


def med(f, k=15):
    f(); f()
    t = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


for n in (4, 20, 32, 64, 128, 256):
    s, t = aligned_copy(sc[:n]), aligned_copy(table[:2 * n])
    G.set_host_thresholds(0, 0)
    gpu_ms = med(lambda: G.pippenger(s, t, n))
    a = G.pippenger(s, t, n)
    G.set_host_thresholds(1 << 20, 64)
    host_ms = med(lambda: G.pippenger(s, t, n), 5)
    assert np.array_equal(a, G.pippenger(s, t, n))
    print("msm n=%4d  gpu %.3f ms  host %.3f ms" % (n, gpu_ms, host_ms), flush=True)
for n in (4, 16, 64):
    x = aligned_copy(sc[:n])
    G.set_host_thresholds(0, 0)
    gpu_ms = med(lambda: G.ntt(x.copy(), "coset_fft"))
    a = G.ntt(x.copy(), "coset_fft")
    G.set_host_thresholds(0, 64)
    host_ms = med(lambda: G.ntt(x.copy(), "coset_fft"))
    assert np.array_equal(a, G.ntt(x.copy(), "coset_fft"))
    print("ntt n=%4d  gpu %.3f ms  host %.3f ms" % (n, gpu_ms, host_ms), flush=True)
