#!/usr/bin/env python3
"""Round-3 additions to the MSM fixtures, from the REFERENCE itself (oracle/_ref/libbbref.so, x86-64 asm path; run in the build container):

  * the sizes around the window-table switch at n = 2^19 (capi.hip add_srs: c = 15 -> 17): 2^19 - 8, 2^19, 2^19 + 8, 3 * 2^18;
  * the skewed scalar sets of bench.skewed_scalars (every scalar equal, {0, 1, -1}, values below 200) at the full 2^20 size;
  * pippenger_low_memory on a PLAIN 1000-point table (test_scalar_multiplication.cpp:164-187).

Inputs are deterministic (splitmix64), so tests/golden/msm_r3.json holds seeds and expected points only.
    python tools/gen_golden_r3.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (inputs only: splitmix64 vectors)
from oracle.pyoracle import Oracle, Ref, aligned_copy  # noqa: E402
from tools.gen_golden import GOLD, SCALAR_SEED, SRS_SEED, digest, hx  # noqa: E402


def main():
    O, R = Oracle(), Ref(True)
    R.set_threads(min(8, os.cpu_count() or 1))
    old = json.load(open(os.path.join(GOLD, "msm.json")))
    x = O.random_scalars(SRS_SEED, 1)[0]
    assert hx(x) == old["srs_secret_mont"]
    n = 1 << 20
    t0 = time.time()
    srs = O.make_srs(x, n)
    assert digest(srs) == old["srs_digest_1048576"]
    table = R.point_table(srs)
    print("srs + table: %.1fs" % (time.time() - t0), flush=True)
    scalars = O.random_scalars(SCALAR_SEED, n)
    out = {"source": "reference scalar_multiplication.cpp batched_scalar_multiplications() / pippenger_low_memory() via oracle/_ref (tools/gen_golden_r3.py)",
           "scalar_seed": "0x%x" % SCALAR_SEED, "srs_seed": "0x%x" % SRS_SEED, "srs_secret_mont": hx(x), "threshold": [], "skewed_2e20": {}}
    # self-check of the recipe: the round-1 fixture at 2^20 through the same call
    chk = R.batched_msm([scalars], [table])[0]
    want = [c for c in old["cases"] if c["n"] == n][0]
    assert hx(chk[0:4]) == want["x"] and hx(chk[4:8]) == want["y"]
    for m in ((1 << 19) - 8, 1 << 19, (1 << 19) + 8, 3 << 18):
        t0 = time.time()
        r = R.batched_msm([aligned_copy(scalars[:m])], [table[:2 * m]])[0]
        print("msm n=%d %.2fs" % (m, time.time() - t0), flush=True)
        out["threshold"].append({"n": m, "x": hx(r[0:4]), "y": hx(r[4:8])})
    for kind in bench.SKEWED_KINDS:
        sc = aligned_copy(bench.skewed_scalars(kind, n))
        t0 = time.time()
        r = R.batched_msm([sc], [table])[0]
        print("skewed %s %.2fs" % (kind, time.time() - t0), flush=True)
        case = {"scalars_sha256": digest(sc)}
        if int(r[7]) >> 63:
            case["infinity"] = True
        else:
            case.update({"x": hx(r[0:4]), "y": hx(r[4:8])})
        # the same set on the first 2^14 points (small enough for the oracle restatement to cross-check on any box)
        r14 = R.batched_msm([aligned_copy(sc[:1 << 14])], [table[:2 << 14]])[0]
        case["first_16384"] = {"x": hx(r14[0:4]), "y": hx(r14[4:8])}
        out["skewed_2e20"][kind] = case
    # pippenger_low_memory: plain table of exactly n * 64 bytes, scalars clobbered by the reference -> a copy goes in
    m = 1000
    plain = aligned_copy(srs[:m])
    r = O.g1_normalize_or_inf(R.pippenger_low_memory(aligned_copy(scalars[:m]), plain, m))
    ref1000 = [c for c in old["cases"] if c["n"] == m and c["forced_bucket_width"] == 0][0]
    assert hx(r[0:4]) == ref1000["x"] and hx(r[4:8]) == ref1000["y"], "pippenger_low_memory(plain points) != pippenger(endo table)"
    out["low_memory_1000"] = {"n": m, "x": hx(r[0:4]), "y": hx(r[4:8]), "note": "equals pippenger() on the endomorphism table of the same points"}
    json.dump(out, open(os.path.join(GOLD, "msm_r3.json"), "w"), indent=0)
    print("wrote", os.path.join(GOLD, "msm_r3.json"))


if __name__ == "__main__":
    main()
