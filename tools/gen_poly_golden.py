#!/usr/bin/env python3
"""tests/golden/poly_ops.json: outputs of the reference's own O(n) polynomial helpers (oracle/_ref/libbbref.so = the
reference sources compiled in place) on small seeded inputs; pins oracle.pyoracle.PolyOracle and, through it, poly.hip.
Run in the build container:  python tools/gen_poly_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import FR_MODULUS, Oracle, Ref, from_int, to_int  # noqa: E402


def hx(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return ["%064x" % (to_int(r) % FR_MODULUS if canonical else to_int(r)) for r in a]


canonical = False


def main():
    global canonical
    O, R = Oracle(), Ref(True)
    out = {"source": "oracle/_ref/libbbref.so through oracle/ref_driver.cpp (see tools/gen_poly_golden.py); inputs = Oracle.random_scalars(seed, n) "
                     "(splitmix64, Montgomery form); values are 256-bit hex of the raw limbs, kate outputs reduced mod r (the reference leaves them coarse)",
           "cases": []}
    for n in (8, 64):
        seed = 0xABCDEF00 + n
        v = O.random_scalars(seed, n)
        z = O.random_scalars(seed + 1, 1)[0]
        w = O.random_scalars(seed + 2, n)
        case = {"n": n, "seed": seed, "z": hx(z)[0]}
        case["evaluate"] = hx(R.evaluate(v, z))[0]
        case["batch_invert"] = hx(R.batch_invert(v))
        canonical = True
        dest, f = R.kate_opening(v, z)
        case["kate_dest"] = hx(dest)
        case["kate_f"] = hx(f)[0]
        canonical = False
        case["pointwise_mul"] = hx(R.pointwise_mul(v, w))
        case["lagrange_l1_fft_2n"] = hx(R.lagrange_l1_fft(n, 2 * n))
        c2, c4 = O.random_scalars(seed + 3, 2 * n), O.random_scalars(seed + 4, 4 * n)
        case["divide_vanishing_2n"] = hx(R.divide_by_pseudo_vanishing(c2, n, 2 * n))
        case["divide_vanishing_4n"] = hx(R.divide_by_pseudo_vanishing(c4, n, 4 * n))
        case["lagrange_evaluations"] = hx(R.lagrange_evaluations(z, n))
        out["cases"].append(case)
    with open(os.path.join(ROOT, "tests", "golden", "poly_ops.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote tests/golden/poly_ops.json")


if __name__ == "__main__":
    main()
