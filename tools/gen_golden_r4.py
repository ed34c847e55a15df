#!/usr/bin/env python3
"""Round-4 additions to the MSM fixtures, from the REFERENCE itself (oracle/_ref/libbbref.so, x86-64 asm path; run in the build container):

  * n = 2^19 + 3: a ragged size (n % 8 != 0) above the two-range switch of the host-pointer entry;
  * n = 2^20 + 8 and n = 2^21: beyond ONE window-table segment (capi.hip add_srs: an SRS above 2^20 points keeps one table per <= 2^20-point
    segment and an MSM over it runs as point-range pieces whose sums are added on the host) -- VERDICT r3 #3;
  * n = 2^20 + 8 on the points [5, 5 + n) of the 2^21-point table: a sub-slice that straddles the segment boundary.

Inputs are deterministic (splitmix64), so tests/golden/msm_r4.json holds seeds and expected points only.
    python tools/gen_golden_r4.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, aligned_copy  # noqa: E402
from tools.gen_golden import GOLD, SCALAR_SEED, SRS_SEED, digest, hx  # noqa: E402


def main():
    O, R = Oracle(), Ref(True)
    R.set_threads(min(8, os.cpu_count() or 1))
    old = json.load(open(os.path.join(GOLD, "msm.json")))
    x = O.random_scalars(SRS_SEED, 1)[0]
    assert hx(x) == old["srs_secret_mont"]
    n = 1 << 21
    t0 = time.time()
    srs = O.make_srs(x, n)
    assert digest(srs[:1 << 20]) == old["srs_digest_1048576"]
    table = R.point_table(srs)
    print("srs + table: %.1fs" % (time.time() - t0), flush=True)
    scalars = O.random_scalars(SCALAR_SEED, n)
    out = {"source": "reference scalar_multiplication.cpp batched_scalar_multiplications() via oracle/_ref (tools/gen_golden_r4.py)",
           "scalar_seed": "0x%x" % SCALAR_SEED, "srs_seed": "0x%x" % SRS_SEED, "srs_secret_mont": hx(x), "srs_digest_2097152": digest(srs), "prefixes": []}
    # self-check of the recipe: the round-1 fixture at 2^20 through the same call
    chk = R.batched_msm([aligned_copy(scalars[:1 << 20])], [table[:2 << 20]])[0]
    want = [c for c in old["cases"] if c["n"] == 1 << 20][0]
    assert hx(chk[0:4]) == want["x"] and hx(chk[4:8]) == want["y"]
    for m in ((1 << 19) + 3, (1 << 20) + 8, 1 << 21):
        t0 = time.time()
        r = R.batched_msm([aligned_copy(scalars[:m])], [table[:2 * m]])[0]
        print("msm n=%d %.2fs" % (m, time.time() - t0), flush=True)
        out["prefixes"].append({"n": m, "x": hx(r[0:4]), "y": hx(r[4:8])})
    off, m = 5, (1 << 20) + 8
    r = R.batched_msm([aligned_copy(scalars[:m])], [aligned_copy(table[2 * off:2 * (off + m)])])[0]
    out["slice"] = {"offset": off, "n": m, "x": hx(r[0:4]), "y": hx(r[4:8]), "note": "scalars[0:n] against points [offset, offset + n)"}
    json.dump(out, open(os.path.join(GOLD, "msm_r4.json"), "w"), indent=0)
    print("wrote", os.path.join(GOLD, "msm_r4.json"))


if __name__ == "__main__":
    main()
