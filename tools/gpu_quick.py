#!/usr/bin/env python3
"""quick GPU bring-up: a few parity checks + timings (not a test, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu
from oracle.pyoracle import Oracle, NTT_KINDS, FR_MODULUS, aligned_copy
from tests.util import noncanonical, SCALAR_SEED, SRS_SEED, NTT_SEED, CONST_SEED

O = Oracle(); G = BbGpu(0)
print(G.version(), "devices", G.device_count(), flush=True)
const = O.random_scalars(CONST_SEED, 1)[0]
for lg in (1, 2, 4, 10, 11, 12, 14):
    n = 1 << lg
    co = noncanonical(O.random_scalars(NTT_SEED + lg, n), FR_MODULUS)
    bad = []
    for kind in NTT_KINDS:
        want = O.ntt(co, kind, const); got = G.ntt(co.copy(), kind, const)
        if not np.array_equal(got, want): bad.append(kind)
    print("ntt 2^%d" % lg, "OK" if not bad else "MISMATCH %s" % bad, flush=True)
for lg in (16, 20, 22):
    n = 1 << lg
    x = O.random_scalars(5 + lg, n)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    for kind in ("fft", "coset_fft", "ifft"):
        G.ntt_device(d.data_ptr(), n, kind); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5): G.ntt_device(d.data_ptr(), n, kind)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print("ntt_device 2^%d %-10s %.3f ms  %.3e elem/s" % (lg, kind, dt * 1e3, n / dt), flush=True)
    y = G.ifft(G.fft(x.copy())); print("roundtrip 2^%d" % lg, np.array_equal(x, y), flush=True)
# MSM
x = O.random_scalars(SRS_SEED, 1)[0]
nsm = 1 << 12
srs = O.make_srs(x, nsm); table = O.point_table(srs); sc = O.random_scalars(SCALAR_SEED, nsm)
for n in (1, 2, 3, 16, 100, 1000, 4096):
    want = O.msm_affine(sc, table, n); got = G.pippenger(sc, table, n)
    print("msm n=%d" % n, "OK" if np.array_equal(got[:8], want[:8]) else "MISMATCH", flush=True)
G.set_timing(True)
for lg in (16, 20):
    n = 1 << lg
    t0 = time.time(); h = G.srs_generate(x, n); print("srs_generate 2^%d %.1f ms" % (lg, (time.time() - t0) * 1e3), flush=True)
    s = O.random_scalars(SCALAR_SEED, n); d = torch.from_numpy(s.view(np.int64)).cuda()
    out = G.msm_device(h, d.data_ptr(), n)
    t0 = time.time()
    for _ in range(3): out = G.msm_device(h, d.data_ptr(), n)
    dt = (time.time() - t0) / 3
    print("msm_device 2^%d %.3f ms wall  %.3e pts/s  stages(ms) %s" % (lg, dt * 1e3, n / dt, ["%.3f" % v for v in G.last_timing()]), flush=True)
    import json
    gold = [c for c in json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/msm.json")))["cases"] if c["n"] == n and "x" in c][0]
    ok = [int(h, 16) for h in gold["x"]] == [int(v) for v in out[:4]]
    print("  golden", "OK" if ok else "MISMATCH", flush=True)
    if not ok: sys.exit(3)
