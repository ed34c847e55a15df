#!/usr/bin/env python3
"""Round-5 additions to the MSM fixtures, from the REFERENCE itself (oracle/_ref/libbbref.so, x86-64 asm path; run in the build container):

  * `straddle`: three jobs of one batched_scalar_multiplications() call, each 2^19 points of the 2^21-point SRS starting at 3 * 2^18 - 11 * k
    (k = 0, 1, 2), i.e. sub-slices that straddle the boundary between the two window-table segments, each against its own scalars
    scalars[k * 2^19 : (k + 1) * 2^19] -- the host-batch path whose helper pieces must wait for the scalars' upload;
  * `rewrite`: 2^16 points of the SRS with ONE point in the middle (index 2^15 + 77) replaced by another SRS point (index 2^16 + 5), scalars[0:2^16]:
    the exact mode of the address-keyed table cache must answer the very next call with this point.

Inputs are deterministic (splitmix64), so tests/golden/msm_r5.json holds seeds and expected points only.
    python tools/gen_golden_r5.py
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
    r4 = json.load(open(os.path.join(GOLD, "msm_r4.json")))
    x = O.random_scalars(SRS_SEED, 1)[0]
    assert hx(x) == r4["srs_secret_mont"]
    n = 1 << 21
    t0 = time.time()
    srs = O.make_srs(x, n)
    assert digest(srs) == r4["srs_digest_2097152"]
    table = R.point_table(srs)
    print("srs + table: %.1fs" % (time.time() - t0), flush=True)
    scalars = O.random_scalars(SCALAR_SEED, n)
    out = {"source": "reference scalar_multiplication.cpp batched_scalar_multiplications() via oracle/_ref (tools/gen_golden_r5.py)",
           "scalar_seed": "0x%x" % SCALAR_SEED, "srs_seed": "0x%x" % SRS_SEED, "srs_secret_mont": hx(x), "srs_digest_2097152": digest(srs)}
    m = 1 << 19
    offs = [3 * (1 << 18) - 11 * k for k in range(3)]
    res = R.batched_msm([aligned_copy(scalars[k * m:(k + 1) * m]) for k in range(3)], [aligned_copy(table[2 * o:2 * (o + m)]) for o in offs])
    out["straddle"] = [{"offset": o, "n": m, "scalars_from": k * m, "x": hx(r[0:4]), "y": hx(r[4:8])} for k, (o, r) in enumerate(zip(offs, res))]
    m, at, src = 1 << 16, (1 << 15) + 77, (1 << 16) + 5
    tab = aligned_copy(table[:2 * m])
    r0 = R.batched_msm([aligned_copy(scalars[:m])], [tab])[0]
    tab[2 * at:2 * at + 2] = table[2 * src:2 * src + 2]
    r1 = R.batched_msm([aligned_copy(scalars[:m])], [tab])[0]
    assert hx(r0[0:4]) != hx(r1[0:4])
    out["rewrite"] = {"n": m, "index": at, "takes_point": src, "before": {"x": hx(r0[0:4]), "y": hx(r0[4:8])}, "after": {"x": hx(r1[0:4]), "y": hx(r1[4:8])}}
    json.dump(out, open(os.path.join(GOLD, "msm_r5.json"), "w"), indent=0)
    print("wrote", os.path.join(GOLD, "msm_r5.json"))


if __name__ == "__main__":
    main()
