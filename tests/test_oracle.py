"""Pins oracle/bn254_oracle.c: (1) the reference tests' own known-answer vectors, (2) fixtures generated from the
reference itself (tools/gen_golden.py -> oracle/_ref), (3) property checks from the reference's test strategy
(test/test_wnaf.cpp, test/test_fr.cpp:239-294, test/test_polynomial_arithmetic.cpp:31-128)."""
import numpy as np
import pytest

from oracle.pyoracle import FQ, FR, FR_MODULUS, NTT_KINDS, aligned_copy, from_int, to_int
from tests.util import CONST_SEED, NTT_SEED, SCALAR_SEED, SRS_SEED, limbs, noncanonical, sha

F = {"fq": FQ, "fr": FR}


def test_reference_field_kats(oracle, golden):
    kats = golden("reference_kats.json")
    for fname in ("fq", "fr"):
        for k in kats[fname]:
            a = limbs(k["a"])
            fn = getattr(oracle, k["op"])
            got = fn(F[fname], a, limbs(k["b"])) if "b" in k else fn(F[fname], a)
            assert np.array_equal(got, limbs(k["expected"])), k["cite"]


def _mont_pt(oracle, d, keys):
    return np.concatenate([oracle.to_mont(FQ, limbs(d[c])) for c in keys])


def _proj_eq(oracle, a, b):
    na, nb = oracle.g1_normalize(a), oracle.g1_normalize(b)
    return np.array_equal(na, nb)


def test_reference_g1_kats(oracle, golden):
    for k in golden("reference_kats.json")["g1"]:
        if k["op"] == "mixed_add":
            got = oracle.g1_mixed_add(_mont_pt(oracle, k["a"], "xyz"), _mont_pt(oracle, k["b"], "xy"))
        elif k["op"] == "dbl3":
            got = _mont_pt(oracle, k["a"], "xyz")
            for _ in range(3):
                got = oracle.g1_dbl(got)
        elif k["op"] == "add":
            got = oracle.g1_add(_mont_pt(oracle, k["a"], "xyz"), _mont_pt(oracle, k["b"], "xyz"))
        else:
            s = oracle.to_mont(FR, limbs(k["scalar"]))
            got = oracle.g1_scalar_mul(oracle.g1_one_affine(), s)
            want = np.concatenate([_mont_pt(oracle, k["expected"], "xy"), oracle.const(FQ, "one")])
            assert np.array_equal(got, want), k["cite"]
            continue
        assert _proj_eq(oracle, got, _mont_pt(oracle, k["expected"], "xyz")), k["cite"]


def test_golden_field_ops(oracle, golden):
    for c in golden("field_ops.json")["cases"]:
        f, a, b = F[c["field"]], limbs(c["a"]), limbs(c["b"])
        for op in ("mul", "mul_coarse", "add", "sub", "add_coarse", "sub_coarse"):
            if op in c:
                assert np.array_equal(getattr(oracle, op)(f, a, b), limbs(c[op])), (c["field"], op)
        for op in ("sqr", "sqr_coarse", "neg", "to_mont", "from_mont", "reduce_once", "invert"):
            if op in c:
                assert np.array_equal(getattr(oracle, op)(f, a), limbs(c[op])), (c["field"], op)


def test_golden_endo_wnaf(oracle, golden):
    lam = oracle.const(FR, "beta")
    for c in golden("endo_wnaf.json")["cases"]:
        k = limbs(c["k"])
        k1, k2 = oracle.split_endo(k)
        assert np.array_equal(k1, limbs(c["k1"])) and np.array_equal(k2, limbs(c["k2"]))
        # k == k1 - lambda*k2 (test_fr.cpp:239-294)
        k1m = oracle.to_mont(FR, np.concatenate([k1, np.zeros(2, dtype=np.uint64)]))
        k2m = oracle.to_mont(FR, np.concatenate([k2, np.zeros(2, dtype=np.uint64)]))
        back = oracle.from_mont(FR, oracle.sub(FR, k1m, oracle.mul(FR, k2m, lam)))
        assert to_int(back) == to_int(k) % FR_MODULUS
        for name, w in c["wnaf"].items():
            half, bits = name.split("_w")
            digits, skew = oracle.fixed_wnaf(k1 if half == "k1" else k2, int(bits))
            assert [int(d) for d in digits] == w["digits"] and skew == w["skew"], name


def test_wnaf_roundtrip(oracle):
    """test_wnaf.cpp:11-131 recover_fixed_wnaf, restated"""
    rng = np.random.default_rng(7)
    cases = [(0, 0), (1, 0), (0, 1)] + [(int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2)), int(rng.integers(0, 1 << 63))) for _ in range(50)]
    for lo, hi in cases:
        for w in (3, 5, 11, 16):
            digits, skew = oracle.fixed_wnaf(np.array([lo, hi], dtype=np.uint64), w)
            entries = len(digits)
            v = 0
            for i, e in enumerate(digits):
                d = ((int(e) & 0x0FFFFFFF) << 1) + 1
                v += (-d if int(e) >> 31 else d) << (w * (entries - 1 - i))
            assert v - skew == lo + (hi << 64)


def test_golden_g1_ops(oracle, golden):
    one = oracle.g1_one_affine()
    for c in golden("g1_ops.json")["cases"]:
        acc = limbs(c["acc"])
        q = limbs(c["scalar_mul_G"])
        assert np.array_equal(oracle.g1_scalar_mul(one, limbs(c["scalar"])), q)
        m = oracle.g1_mixed_add(acc, q[:8])
        assert np.array_equal(m, limbs(c["mixed_add"]))
        a = oracle.g1_add(m, acc)
        assert np.array_equal(a, limbs(c["add"]))
        d = oracle.g1_dbl(a)
        assert np.array_equal(d, limbs(c["dbl"]))
        assert np.array_equal(oracle.g1_normalize(d), limbs(c["normalize"]))


def test_g1_exception_cases(oracle):
    """test_g1.cpp:124-241: P+(-P)=inf, P+P=dbl, inf+P=P"""
    one = oracle.g1_one_affine()
    s = oracle.random_scalars(99, 1)[0]
    p = oracle.g1_scalar_mul(one, s)
    neg = p.copy()
    neg[4:8] = oracle.neg(FQ, p[4:8])
    assert oracle.is_infinity(oracle.g1_add(p, neg))
    assert oracle.is_infinity(oracle.g1_mixed_add(p, neg[:8]))
    assert np.array_equal(oracle.g1_normalize(oracle.g1_add(p, p)), oracle.g1_normalize(oracle.g1_dbl(p)))
    assert np.array_equal(oracle.g1_normalize(oracle.g1_mixed_add(p, p[:8])), oracle.g1_normalize(oracle.g1_dbl(p)))
    inf = np.zeros(12, dtype=np.uint64)
    inf[7] = np.uint64(1 << 63)
    assert np.array_equal(oracle.g1_add(inf, p), p) and np.array_equal(oracle.g1_add(p, inf), p)
    assert np.array_equal(oracle.g1_mixed_add(inf, p[:8])[:8], p[:8])
    assert oracle.is_infinity(oracle.g1_dbl(inf))


@pytest.fixture(scope="module")
def msm_inputs(oracle, golden):
    g = golden("msm.json")
    x = limbs(g["srs_secret_mont"])
    assert np.array_equal(x, oracle.random_scalars(SRS_SEED, 1)[0])
    n = 4096
    srs = oracle.make_srs(x, n)
    for i, v in g["srs_samples"].items():
        assert np.array_equal(srs[int(i)], limbs(v))
    table = oracle.point_table(srs)
    assert sha(table[:8192]) == g["table_digest_4096"]
    scalars = oracle.random_scalars(SCALAR_SEED, n)
    return g, srs, table, scalars


def test_bucket_width_table(oracle):
    for n, c in ((1 << 20, 15), (1 << 16, 12), (131072, 15), (8192, 10), (10000, 10), (1, 1), (0, 1), (2, 2), (99999, 12), (100000, 15)):
        assert oracle.optimal_bucket_width(n) == c


def test_golden_msm(oracle, msm_inputs):
    g, srs, table, scalars = msm_inputs
    done = 0
    for c in g["cases"]:
        n = c["n"]
        if n > 4096 or "x" not in c or "scalars" in c or "points" in c:
            continue
        out = oracle.msm_affine(scalars, table, n, c["forced_bucket_width"])
        assert np.array_equal(out[0:4], limbs(c["x"])) and np.array_equal(out[4:8], limbs(c["y"])), c
        done += 1
    assert done >= 8


def test_golden_msm_edge_cases(oracle, msm_inputs):
    from oracle.pyoracle import aligned_copy
    g, srs, table, scalars = msm_inputs
    seen = 0
    for c in g["cases"]:
        if c.get("scalars") == "even-index scalars zero":
            zs = scalars[:64].copy()
            zs[::2] = 0
            out = oracle.msm_affine(aligned_copy(zs), table, 64)
        elif c.get("scalars") == "all zero":
            out = oracle.pippenger(aligned_copy(np.zeros((16, 4), dtype=np.uint64)), table, 16)
            assert oracle.is_infinity(out) == c["infinity"]
            seen += 1
            continue
        elif c["n"] == 0:
            assert oracle.is_infinity(oracle.pippenger(scalars, table, 0)) == c["infinity"]
            seen += 1
            continue
        elif "points" in c:
            same_t = oracle.point_table(aligned_copy(np.tile(srs[5], (32, 1))))
            sc = aligned_copy(np.tile(oracle.const(FR, "one"), (32, 1))) if c.get("scalars") == "all one" else scalars
            out = oracle.msm_affine(sc, same_t, 32)
        else:
            continue
        assert np.array_equal(out[0:4], limbs(c["x"])) and np.array_equal(out[4:8], limbs(c["y"])), c
        seen += 1
    assert seen == 5


def test_golden_batched_msm(oracle, msm_inputs):
    """batched_scalar_multiplications (scalar_multiplication.cpp:650-772): normalised outputs, any thread split"""
    import ctypes as C
    from oracle.pyoracle import aligned_copy, ptr
    g, srs, table, scalars = msm_inputs
    big = oracle.random_scalars(SCALAR_SEED, 3 * 4096)

    class Job(C.Structure):
        _fields_ = [("points", C.POINTER(C.c_uint64)), ("scalars", C.POINTER(C.c_uint64)), ("n", C.c_size_t), ("out", C.c_uint64 * 12)]
    for threads in (1, 8):
        keep = [aligned_copy(big[o:o + 4096]) for o in (0, 4096, 8192)]
        jobs = (Job * 3)()
        for j, s in zip(jobs, keep):
            j.points, j.scalars, j.n = ptr(table), ptr(s), 4096
        assert oracle.lib.orc_batched_msm(jobs, C.c_size_t(3), C.c_size_t(threads)) == 0
        for j, want in zip(jobs, g["batched_3x4096"]):
            got = np.array(list(j.out), dtype=np.uint64)
            assert np.array_equal(got[0:4], limbs(want["x"])) and np.array_equal(got[4:8], limbs(want["y"]))
            assert np.array_equal(got[8:12], limbs(want["z"]))


def test_msm_vs_naive(oracle, msm_inputs):
    """test_scalar_multiplication.cpp:72-104 shape: pippenger == sum of individual scalar multiplications"""
    g, srs, table, scalars = msm_inputs
    n = 40
    acc = np.zeros(12, dtype=np.uint64)
    acc[7] = np.uint64(1 << 63)
    for i in range(n):
        acc = oracle.g1_add(acc, oracle.g1_scalar_mul(srs[i], scalars[i]))
    assert np.array_equal(oracle.g1_normalize(acc), oracle.msm_affine(scalars, table, n))


def test_golden_ntt_small(oracle, golden):
    g = golden("ntt.json")
    const = limbs(g["constant"])
    assert np.array_equal(const, oracle.random_scalars(CONST_SEED, 1)[0])
    for c in g["small"]:
        co = limbs(c["input"]).reshape(-1, 4)
        assert np.array_equal(co, noncanonical(oracle.random_scalars(NTT_SEED, c["n"]), FR_MODULUS))
        out = oracle.ntt(co, c["kind"], const)
        assert np.array_equal(out.reshape(-1), limbs(c["output"])), (c["n"], c["kind"])


def test_golden_ntt_digests(oracle, golden):
    g = golden("ntt.json")
    const = limbs(g["constant"])
    for c in g["large"]:
        n = c["n"]
        if n > 4096:
            continue
        co = noncanonical(oracle.random_scalars(NTT_SEED, n), FR_MODULUS)
        out = oracle.ntt(co, c["kind"], const)
        assert sha(out) == c["sha256"], (n, c["kind"])
        for i, v in c["samples"].items():
            assert np.array_equal(out[int(i)], limbs(v))


def test_golden_ntt_2_16_fft(oracle, golden):
    g = golden("ntt.json")
    c = [c for c in g["large"] if c["n"] == 65536 and c["kind"] == "coset_fft"][0]
    co = noncanonical(oracle.random_scalars(NTT_SEED, 65536), FR_MODULUS)
    assert sha(oracle.ntt(co, "coset_fft")) == c["sha256"]


def test_fft_matches_direct_evaluation(oracle):
    """test_polynomial_arithmetic.cpp:31-56"""
    n = 16
    poly = oracle.random_scalars(4242, n)
    out = oracle.ntt(poly, "fft")
    root = oracle.root_of_unity(4)
    w = oracle.const(FR, "one")
    for i in range(n):
        assert np.array_equal(out[i], oracle.evaluate(poly, w))
        w = oracle.mul(FR, w, root)


def test_fft_roundtrips(oracle):
    """test_polynomial_arithmetic.cpp:58-128"""
    for n in (2, 256, 1024):
        poly = oracle.random_scalars(31337 + n, n)
        canon = np.stack([oracle.reduce_once(FR, p) for p in poly])
        assert np.array_equal(oracle.ntt(oracle.ntt(poly, "fft"), "ifft"), canon)
        assert np.array_equal(oracle.ntt(oracle.ntt(poly, "coset_fft"), "coset_ifft"), canon)


def test_ntt_kinds_complete():
    assert set(NTT_KINDS) == {"fft", "ifft", "coset_fft", "coset_ifft", "fft_with_constant", "ifft_with_constant",
                              "coset_fft_with_constant"}


# ---- the O(n) polynomial helpers: PolyOracle pinned by fixtures made by the reference itself -----------------
def _hx(a, canonical=False):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return ["%064x" % (to_int(r) % FR_MODULUS if canonical else to_int(r)) for r in a]


def test_poly_oracle_matches_reference_fixtures(oracle, golden):
    from oracle.pyoracle import PolyOracle as P
    for case in golden("poly_ops.json")["cases"]:
        n, seed = case["n"], case["seed"]
        v, z, w = oracle.random_scalars(seed, n), oracle.random_scalars(seed + 1, 1)[0], oracle.random_scalars(seed + 2, n)
        assert _hx(z)[0] == case["z"]
        assert _hx(P.evaluate(v, z))[0] == case["evaluate"]
        assert _hx(oracle.evaluate(v, z))[0] == case["evaluate"]
        assert _hx(P.batch_invert(v)) == case["batch_invert"]
        dest, f = P.kate_opening(v, z)
        assert _hx(dest) == case["kate_dest"] and _hx(f)[0] == case["kate_f"]
        assert _hx(P.pointwise_mul(v, w)) == case["pointwise_mul"]
        assert _hx(P.lagrange_l1_fft(n, 2 * n)) == case["lagrange_l1_fft_2n"]
        c2, c4 = oracle.random_scalars(seed + 3, 2 * n), oracle.random_scalars(seed + 4, 4 * n)
        assert _hx(P.divide_by_pseudo_vanishing(c2, n, 2 * n)) == case["divide_vanishing_2n"]
        assert _hx(P.divide_by_pseudo_vanishing(c4, n, 4 * n)) == case["divide_vanishing_4n"]


def test_poly_oracle_internal_consistency(oracle):
    """properties the restatement must have whatever the reference does: inverse * value = 1, (X - z) W(X) + F(z) = F(X),
    the exclusive prefix product is the reference's accumulator chain (prover.cpp:194-202), sigma of the identity mapping"""
    from oracle.pyoracle import PolyOracle as P
    n = 32
    v, z = oracle.random_scalars(77, n), oracle.random_scalars(78, 1)[0]
    one = P.mont([1])[0]
    inv = P.batch_invert(v)
    assert all(np.array_equal(r, one) for r in P.pointwise_mul(v, inv))
    dest, f = P.kate_opening(v, z)
    fp, wp, zp, fz = P.plain(v), P.plain(dest), P.plain(z)[0], P.plain(f)[0]
    assert wp[n - 1] == 0
    for i in range(n):
        lhs = ((wp[i - 1] if i else 0) - zp * wp[i] + (fz if i == 0 else 0)) % FR_MODULUS
        assert lhs == fp[i]
    ex = P.plain(P.product_scan(v))
    chain = [1]
    for a in P.plain(v)[:-1]:
        chain.append(chain[-1] * a % FR_MODULUS)
    assert ex == chain
    assert P.plain(P.product_scan(v, reverse=True, inclusive=True))[0] == chain[-1] * P.plain(v)[-1] % FR_MODULUS
    ident = np.arange(n, dtype=np.uint32)
    w = P.root(5)
    assert P.plain(P.permutation_lagrange_base(ident, n)) == [pow(w, i, FR_MODULUS) for i in range(n)]
    assert P.plain(P.permutation_lagrange_base(ident + np.uint32(1 << 31), n)) == [7 * pow(w, i, FR_MODULUS) % FR_MODULUS for i in range(n)]


def test_oracle_on_skewed_scalars_and_plain_tables(oracle, golden):
    """round-3 fixtures (tests/golden/msm_r3.json, outputs of the reference: tools/gen_golden_r3.py): bench.skewed_scalars reproduces the vectors
    the reference was run on (SHA-256), and the restatement gives the reference's points for their 2^14-point prefixes and for the
    pippenger_low_memory case (scalar_multiplication.cpp:142-262: same sum as pippenger on the endomorphism table)"""
    import bench
    from tests.util import SCALAR_SEED, limbs, sha
    g = golden("msm_r3.json")
    m = 1 << 14
    srs = oracle.make_srs(limbs(g["srs_secret_mont"]), m)
    table = oracle.point_table(srs)
    for kind in bench.SKEWED_KINDS:
        case = g["skewed_2e20"][kind]
        sc = bench.skewed_scalars(kind, 1 << 20)
        assert sha(sc) == case["scalars_sha256"], kind
        got = oracle.msm_affine(aligned_copy(sc[:m]), table, m)
        assert np.array_equal(got[0:4], limbs(case["first_16384"]["x"])) and np.array_equal(got[4:8], limbs(case["first_16384"]["y"])), kind
    case = g["low_memory_1000"]
    got = oracle.msm_affine(oracle.random_scalars(SCALAR_SEED, case["n"]), table, case["n"])
    assert np.array_equal(got[0:4], limbs(case["x"])) and np.array_equal(got[4:8], limbs(case["y"]))
