#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE itself (oracle/_ref/libbbref.so = the reference's own
scalar_multiplication.cpp / polynomial_arithmetic.cpp / evaluation_domain.cpp compiled in place, x86-64 asm path).

Run in the build container (needs /root/reference):   python tools/gen_golden.py [--big]
Inputs are deterministic (splitmix64 streams) so the fixtures hold only seeds + expected outputs
(or SHA-256 digests + sampled elements for large vectors).  `--big` adds the 2^20 MSM and 2^20 / 2^22 NTT cases.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import (FQ, FR, FR_MODULUS, NTT_KINDS, Oracle, Ref, aligned_copy, from_int, to_int)  # noqa: E402

SCALAR_SEED = 0x9E3779B97F4A7C15
SRS_SEED = 0x5EED0F5EC2E7C0DE
NTT_SEED = 0x0123456789ABCDEF
CONST_SEED = 0x00C0FFEE00C0FFEE
GOLD = os.path.join(ROOT, "tests", "golden")


def hx(a):
    return ["0x%016x" % int(v) for v in np.asarray(a, dtype=np.uint64).reshape(-1)]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def noncanonical(coeffs):
    """every third element gets +r so inputs cover [0, 2r)"""
    out = coeffs.copy()
    for i in range(0, out.shape[0], 3):
        out[i] = from_int(to_int(out[i]) + FR_MODULUS)
    return out


def noncanonical_fast(coeffs):
    out = coeffs.copy()
    mod = from_int(FR_MODULUS)
    idx = np.arange(0, out.shape[0], 3)
    carry = np.zeros(idx.shape[0], dtype=np.uint64)
    for l in range(4):
        a = out[idx, l]
        s = a + mod[l]
        c1 = (s < a).astype(np.uint64)
        s2 = s + carry
        c2 = (s2 < s).astype(np.uint64)
        out[idx, l] = s2
        carry = c1 + c2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    O, R = Oracle(), Ref(True)
    rng_state = [0xA5A5A5A5DEADBEEF]

    def rnd256(bits=256):
        import ctypes as C
        v = 0
        st = C.c_uint64(rng_state[0])
        for i in range(4):
            v |= int(O.lib.orc_splitmix64(C.byref(st))) << (64 * i)
        rng_state[0] = st.value
        return from_int(v & ((1 << bits) - 1))

    # ---------------- field ops (reference asm path; operands < 2p as on the prover path, plus canonical) -------------
    field = []
    for f, name, mod in ((FQ, "fq", None), (FR, "fr", None)):
        modulus = to_int(O.const(f, "modulus"))
        for it in range(24):
            a = from_int(to_int(rnd256()) % (2 * modulus if it % 2 else modulus))
            b = from_int(to_int(rnd256()) % (2 * modulus if it % 4 == 1 else modulus))
            case = {"field": name, "a": hx(a), "b": hx(b)}
            for op in ("mul", "sqr", "mul_coarse", "sqr_coarse"):
                case[op] = hx(R.field_op(f, op, a, b))
            if it % 2 == 0:  # canonical operands: add/sub/neg/montgomery/invert are defined identically on both paths
                for op in ("add", "sub", "neg", "to_mont", "from_mont", "add_coarse", "sub_coarse", "reduce_once"):
                    case[op] = hx(R.field_op(f, op, a, b))
                if it % 8 == 0:
                    case["invert"] = hx(R.field_op(f, "invert", a))
            field.append(case)
    json.dump({"source": "oracle/_ref/libbbref.so (reference asm path)", "cases": field},
              open(os.path.join(GOLD, "field_ops.json"), "w"), indent=0)

    # ---------------- endo split + wNAF ------------------------------------------------------------------------------
    endo = []
    specials = [0, 1, 2, FR_MODULUS - 1, FR_MODULUS - 2, (1 << 128) - 1, 1 << 127, (1 << 253) + 5]
    for i in range(64):
        k = specials[i] if i < len(specials) else to_int(rnd256()) % FR_MODULUS
        kk = from_int(k)
        k1, k2 = R.split_endo(kk)
        case = {"k": hx(kk), "k1": hx(k1), "k2": hx(k2), "wnaf": {}}
        for w in (2, 5, 10, 13, 16):
            for nm, s in (("k1", k1), ("k2", k2)):
                if int(s[1]) >> 63:
                    continue
                dig, skew = R.fixed_wnaf(s, w)
                case["wnaf"]["%s_w%d" % (nm, w)] = {"digits": [int(d) for d in dig], "skew": skew}
        endo.append(case)
    json.dump({"source": "reference field.hpp:413-485, wnaf.hpp:38-55 via oracle/_ref", "cases": endo},
              open(os.path.join(GOLD, "endo_wnaf.json"), "w"), indent=0)

    # ---------------- group ops --------------------------------------------------------------------------------------
    one = O.g1_one_affine()
    fq_one = O.const(FQ, "one")
    group = []
    P = np.concatenate([one, fq_one])
    acc = R.g1_op("dbl", P)
    for i in range(12):
        s = O.random_scalars(1000 + i, 1)[0]
        q = R.g1_scalar_mul(one, s)  # affine (z=one)
        m = R.g1_op("mixed_add", acc, q[:8])
        a = R.g1_op("add", m, acc)
        d = R.g1_op("dbl", a)
        nrm = R.g1_op("normalize", d)
        group.append({"scalar": hx(s), "scalar_mul_G": hx(q), "acc": hx(acc), "mixed_add": hx(m), "add": hx(a),
                      "dbl": hx(d), "normalize": hx(nrm)})
        acc = d
    json.dump({"source": "reference group.hpp via oracle/_ref", "cases": group},
              open(os.path.join(GOLD, "g1_ops.json"), "w"), indent=0)

    # ---------------- MSM --------------------------------------------------------------------------------------------
    x = O.random_scalars(SRS_SEED, 1)[0]
    nmax = (1 << 20) if args.big else (1 << 16)
    t0 = time.time()
    srs = O.make_srs(x, nmax)
    print("srs %d points: %.1fs" % (nmax, time.time() - t0))
    # pin the synthetic SRS against the reference's own scalar multiplication at sampled indices
    srs_samples = {}
    xp = O.const(FR, "one")
    pw = {}
    cur = xp
    for i in range(0, 9):
        pw[i] = cur
        cur = O.mul(FR, cur, x)
    for i in (0, 1, 2, 3, 8):
        want = R.g1_scalar_mul(one, pw[i])[:8]
        assert np.array_equal(want, srs[i]), i
        srs_samples[str(i)] = hx(srs[i])
    table = R.point_table(srs)
    assert np.array_equal(table, O.point_table(srs))
    scalars = O.random_scalars(SCALAR_SEED, nmax)
    msm = {"source": "reference scalar_multiplication.cpp pippenger()/batched_scalar_multiplications() via oracle/_ref",
           "scalar_seed": "0x%x" % SCALAR_SEED, "srs_seed": "0x%x" % SRS_SEED, "srs_secret_mont": hx(x),
           "srs_samples": srs_samples, "srs_digest_65536": digest(srs[:65536]),
           "table_digest_4096": digest(table[:8192]), "cases": []}
    sizes = [(1, 0), (2, 0), (3, 0), (3, 5), (16, 0), (100, 7), (1000, 0), (4096, 0), (4096, 12), (10000, 0), (65536, 0),
             (65536, 12), (65536, 15)]
    if args.big:
        msm["srs_digest_1048576"] = digest(srs)
        sizes += [(1 << 20, 0)]
    for n, c in sizes:
        t0 = time.time()
        if n >= 65536:
            out = R.batched_msm([scalars[:n]], [table[:2 * n]])[0]
        else:
            out = O.g1_normalize_or_inf(R.pippenger(scalars, table, n, c))
        print("msm n=%d c=%d %.2fs" % (n, c, time.time() - t0))
        msm["cases"].append({"n": n, "forced_bucket_width": c, "x": hx(out[0:4]), "y": hx(out[4:8])})
    # structured scalar sets (edge cases the reference tests: zero scalar, n=0, repeated points)
    zs = scalars[:64].copy()
    zs[::2] = 0
    out = O.g1_normalize_or_inf(R.pippenger(aligned_copy(zs), table, 64, 0))
    msm["cases"].append({"n": 64, "forced_bucket_width": 0, "scalars": "even-index scalars zero", "x": hx(out[0:4]), "y": hx(out[4:8])})
    allzero = aligned_copy(np.zeros((16, 4), dtype=np.uint64))
    out = R.pippenger(allzero, table, 16, 0)
    msm["cases"].append({"n": 16, "forced_bucket_width": 0, "scalars": "all zero", "infinity": bool(int(out[7]) >> 63)})
    out = R.pippenger(allzero, table, 0, 0)
    msm["cases"].append({"n": 0, "forced_bucket_width": 0, "infinity": bool(int(out[7]) >> 63)})
    # all points equal (forces the P+P doubling branch inside bucket accumulation)
    same = aligned_copy(np.tile(srs[5], (32, 1)))
    same_t = R.point_table(same)
    out = O.g1_normalize_or_inf(R.pippenger(scalars, same_t, 32, 0))
    msm["cases"].append({"n": 32, "forced_bucket_width": 0, "points": "all equal to srs[5]", "x": hx(out[0:4]), "y": hx(out[4:8])})
    ones = aligned_copy(np.tile(O.const(FR, "one"), (32, 1)))
    out = O.g1_normalize_or_inf(R.pippenger(ones, same_t, 32, 0))
    msm["cases"].append({"n": 32, "forced_bucket_width": 0, "points": "all equal to srs[5]", "scalars": "all one",
                         "x": hx(out[0:4]), "y": hx(out[4:8])})
    # batched: 3 jobs over shifted scalar windows (prover shape, prover.cpp:65-86)
    jobs = [aligned_copy(scalars[o:o + 4096]) for o in (0, 4096, 8192)]
    outs = R.batched_msm(jobs, [table[:8192]] * 3)
    msm["batched_3x4096"] = [{"offset": o, "x": hx(v[0:4]), "y": hx(v[4:8]), "z": hx(v[8:12])} for o, v in zip((0, 4096, 8192), outs)]
    json.dump(msm, open(os.path.join(GOLD, "msm.json"), "w"), indent=0)

    # ---------------- NTT --------------------------------------------------------------------------------------------
    const = O.random_scalars(CONST_SEED, 1)[0]
    ntt = {"source": "reference polynomial_arithmetic.cpp fft family via oracle/_ref", "seed": "0x%x" % NTT_SEED,
           "constant_seed": "0x%x" % CONST_SEED, "constant": hx(const),
           "input": "orc_random_scalars(seed, n) with +r added to every third element (non-canonical [0,2r) inputs)",
           "small": [], "large": []}
    for lg in (1, 2, 3, 4):
        n = 1 << lg
        co = noncanonical(O.random_scalars(NTT_SEED, n))
        for kind in NTT_KINDS:
            ntt["small"].append({"n": n, "kind": kind, "input": hx(co), "output": hx(R.ntt(co, kind, const))})
    big_logs = [8, 10, 12, 16] + ([20, 22] if args.big else [])
    for lg in big_logs:
        n = 1 << lg
        base = O.random_scalars(NTT_SEED, n)
        co = noncanonical_fast(base)
        if lg == 8:
            assert np.array_equal(co, noncanonical(base))
        R.prepare_domain(n)
        for kind in NTT_KINDS:
            t0 = time.time()
            out = R.ntt(co, kind, const)
            samples = {str(i): hx(out[i]) for i in (0, 1, 2, n // 2 - 1, n // 2, n - 2, n - 1)}
            ntt["large"].append({"n": n, "kind": kind, "sha256": digest(out), "samples": samples})
            print("ntt 2^%d %s %.2fs" % (lg, kind, time.time() - t0))
    json.dump(ntt, open(os.path.join(GOLD, "ntt.json"), "w"), indent=0)
    print("wrote fixtures to", GOLD)


if __name__ == "__main__":
    main()
