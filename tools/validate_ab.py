#!/usr/bin/env python3
"""cost of the EXACT mode of the address-keyed point-table cache (bbgpu_srs_set_validate / BBGPU_SRS_VALIDATE=full) at the host-pointer entry:
bbgpu_msm_g1 at 2^16 and 2^20 points against a cached table, 16 sampled rows per call (default) against every row per call; median of 15 after 5,
alternating, and the batched entry with three jobs over the same table (one check per distinct range)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu
G = BbGpu(0)
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)


def med(f, reps=15, warm=5):
    for _ in range(warm): f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for lg in (16, 20):
    n = 1 << lg
    h0, table = G.srs_generate(x, n, True)
    G.srs_release(h0)
    hs = []
    for k in range(3):
        sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        hs.append(sc)
    turn = [0]

    def one():
        turn[0] += 1
        return G.pippenger(hs[turn[0] % 3], table, n)

    def batch():
        return G.batched_scalar_multiplications([(table, hs[k], n) for k in range(3)])
    ref = one()
    h = G.srs_register(table)  # the cached copy the host-pointer calls are served from
    out = {}
    for rnd in range(2):
        for full in (0, 1):
            G.srs_set_validate(h, full)
            out.setdefault(("msm", full), []).append(med(one))
            out.setdefault(("batch3", full), []).append(med(batch, 7, 2))
    assert np.array_equal(one(), ref)
    G.srs_set_validate(h, 0)
    for what in ("msm", "batch3"):
        s, f = out[(what, 0)], out[(what, 1)]
        print("2^%d  %-28s sampled %s ms   full %s ms   (+%.1f %%)" % (lg, "bbgpu_msm_g1" if what == "msm" else "bbgpu_msm_g1_batch, 3 jobs", " / ".join("%.3f" % v for v in s),
                                                                      " / ".join("%.3f" % v for v in f), (min(f) / min(s) - 1) * 100), flush=True)
    G.srs_release(h)
G.shutdown()
