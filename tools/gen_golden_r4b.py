#!/usr/bin/env python3
"""Round-4 fixtures above the sizes the earlier ones cover, from the REFERENCE itself (oracle/_ref/libbbref.so, x86-64 asm path; run in the
build container, ~5 minutes and ~3 GiB):

  * MSM over prefixes of a 2^22-point SRS: n = 2^22 (four window-table segments, capi.hip add_srs) and n = 3 * 2^20 + 11 (ragged, the last
    segment partly used);
  * all seven transform kinds at 2^23 and 2^24 (the sizes the extended 4n coset domain of a 2^21 / 2^22-gate circuit needs; 2^24 is past
    the point where ntt.hip's second pass runs as two launches): SHA-256 of the output + sampled elements.

Inputs are deterministic (splitmix64), so tests/golden/big_r4.json holds seeds, digests and expected points only.
    python tools/gen_golden_r4b.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import NTT_KINDS, Oracle, Ref, aligned_copy  # noqa: E402
from tools.gen_golden import CONST_SEED, GOLD, NTT_SEED, SCALAR_SEED, SRS_SEED, digest, hx, noncanonical_fast  # noqa: E402


def main():
    O, R = Oracle(), Ref(True)
    R.set_threads(min(8, os.cpu_count() or 1))
    old = json.load(open(os.path.join(GOLD, "msm_r4.json")))
    x = O.random_scalars(SRS_SEED, 1)[0]
    assert hx(x) == old["srs_secret_mont"]
    out = {"source": "reference scalar_multiplication.cpp batched_scalar_multiplications() and polynomial_arithmetic.cpp fft family via oracle/_ref "
                     "(tools/gen_golden_r4b.py)",
           "scalar_seed": "0x%x" % SCALAR_SEED, "srs_seed": "0x%x" % SRS_SEED, "srs_secret_mont": hx(x), "ntt_seed": "0x%x" % NTT_SEED,
           "constant_seed": "0x%x" % CONST_SEED, "msm": [], "ntt": []}
    n = 1 << 22
    t0 = time.time()
    srs = O.make_srs(x, n)
    assert digest(srs[:1 << 21]) == old["srs_digest_2097152"]
    out["srs_digest_%d" % n] = digest(srs)
    table = R.point_table(srs)
    del srs
    print("srs + table: %.1fs" % (time.time() - t0), flush=True)
    scalars = O.random_scalars(SCALAR_SEED, n)
    chk = R.batched_msm([aligned_copy(scalars[:1 << 21])], [table[:2 << 21]])[0]  # self-check of the recipe against the 2^21 fixture
    want = [c for c in old["prefixes"] if c["n"] == 1 << 21][0]
    assert hx(chk[0:4]) == want["x"] and hx(chk[4:8]) == want["y"]
    for m in (3 * (1 << 20) + 11, n):
        t0 = time.time()
        r = R.batched_msm([aligned_copy(scalars[:m])], [table[:2 * m]])[0]
        print("msm n=%d %.2fs" % (m, time.time() - t0), flush=True)
        out["msm"].append({"n": m, "x": hx(r[0:4]), "y": hx(r[4:8])})
    del table, scalars
    const = O.random_scalars(CONST_SEED, 1)[0]
    out["constant"] = hx(const)
    for lg in (23, 24):
        n = 1 << lg
        co = noncanonical_fast(O.random_scalars(NTT_SEED, n))
        R.prepare_domain(n)
        for kind in NTT_KINDS:
            t0 = time.time()
            res = R.ntt(co, kind, const)
            samples = {str(i): hx(res[i]) for i in (0, 1, 2, n // 2 - 1, n // 2, n - 2, n - 1)}
            out["ntt"].append({"n": n, "kind": kind, "sha256": digest(res), "samples": samples})
            print("ntt 2^%d %s %.2fs" % (lg, kind, time.time() - t0), flush=True)
            del res
    json.dump(out, open(os.path.join(GOLD, "big_r4.json"), "w"), indent=0)
    print("wrote", os.path.join(GOLD, "big_r4.json"))


if __name__ == "__main__":
    main()
