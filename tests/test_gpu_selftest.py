"""Known-answer tests of the DEVICE field and group layer (csrc/fe.hpp with the gfx950 asm products, csrc/g1.hpp) through
bbgpu_selftest_field / bbgpu_selftest_g1: the reference tests' own vectors (tests/golden/reference_kats.json <-
test/test_fq.cpp:51-133, test_fr.cpp:51-88, test_g1.cpp:41-122), outputs of the reference itself (field_ops.json, g1_ops.json)
and the lazy-bound extremes of the 9 x 29-bit representation against exact integer arithmetic.  Bit-exact (integer work)."""
import numpy as np
import pytest

from oracle.pyoracle import FQ, FR, FQ_MODULUS, FR_MODULUS, from_int, to_int
from tests.util import limbs

pytestmark = pytest.mark.gpu

MOD = {"fq": FQ_MODULUS, "fr": FR_MODULUS}
R256 = 1 << 256


@pytest.fixture(scope="module")
def gpu():
    from barretenberg_amd import BbGpu
    g = BbGpu(device=0)
    yield g
    g.shutdown()


def _ints(arr):
    return [to_int(r) for r in arr]


def _expected(op, a, b, p):
    """value the op names, on Montgomery-2^256 residues: a~ = a R.  Products return (x y) R, i.e. a~ b~ / R."""
    rinv = pow(R256, -1, p)
    mm = lambda x, y: x * y * rinv % p
    return {
        "mul": mm(a, b), "sqr": mm(a, a), "add": (a + b) % p, "sub": (a - b) % p, "neg": (-a) % p,
        "mul_add": (mm(a, b) + mm(a + b, a - b)) % p, "mul_sub": (mm(a, b) - mm(2 * a, b)) % p,
        "lazy_limbs": mm(2 * a, 3 * b), "lazy_weak": mm(4 * a, b - a), "lazy_value": 28 * a % p, "reduce": 28 * a % p,
        "sqr_lazy": mm(2 * a - b, 2 * a - b),
        "mul_addhi": (mm(a, b) - a) % p, "sqr_addhi": (mm(a, a) - b - 2 * a) % p,
    }[op]


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_device_field_ops_vs_reference_outputs(gpu, golden, field):
    """field_ops.json: outputs of the reference's own asm path on the same operands"""
    p = MOD[field]
    cases = [c for c in golden("field_ops.json")["cases"] if c["field"] == field]
    a = np.stack([limbs(c["a"]) for c in cases])
    b = np.stack([limbs(c["b"]) for c in cases])
    for op in ("mul", "sqr", "add", "sub", "neg"):
        got = _ints(gpu.selftest_field(field, op, a, b))
        hits = 0
        for c, g in zip(cases, got):
            if op in c:  # not every case carries every op
                assert g == to_int(limbs(c[op])) % p and g < p, (field, op, c["a"])
                hits += 1
        assert hits >= 4, (field, op)


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_device_field_reference_kats(gpu, golden, field):
    """the hard-coded vectors of test_fq.cpp / test_fr.cpp, including their out-of-range raw operands (any 256-bit value is a
    legal input of the device layer; the device answers with the canonical representative)"""
    p = MOD[field]
    done = 0
    for k in golden("reference_kats.json")[field]:
        if k["op"] not in ("mul", "sqr", "add", "sub"):
            continue
        a = limbs(k["a"]).reshape(1, 4)
        b = limbs(k["b"]).reshape(1, 4) if "b" in k else a
        got = to_int(gpu.selftest_field(field, k["op"], a, b)[0])
        assert got == to_int(limbs(k["expected"])) % p, k["cite"]
        done += 1
    assert done >= 3


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_device_field_lazy_bounds(gpu, oracle, field):
    """every op on random and on extreme operands (0, 1, p - 1, p, 2p - 1, 2^256 - 1, all limbs 2^29 - 1 ...) against exact
    integers: unnormalised limbs at the multiplier's limit (L1 L2 = 6), beyond it (renormalised), values up to 168 p"""
    p = MOD[field]
    rng = np.random.default_rng(20261004)
    special = [0, 1, 2, p - 1, p, p + 1, 2 * p - 1, 2 * p, 5 * p, R256 - 1, R256 - 2, (1 << 255), (1 << 232) - 1,
               sum(((1 << 29) - 1) << (29 * i) for i in range(8)), sum(1 << (29 * i) for i in range(9)) % R256, 0x1FFFFFFF, 1 << 29]
    vals = special + [int.from_bytes(rng.bytes(32), "little") for _ in range(200)]
    pairs = [(x, y) for x in special for y in special[:9]] + list(zip(vals, reversed(vals))) + [(v, v) for v in vals[:40]]
    a = np.stack([from_int(x) for x, _ in pairs])
    b = np.stack([from_int(y) for _, y in pairs])
    for op in ("mul", "sqr", "add", "sub", "neg", "mul_add", "mul_sub", "lazy_limbs", "lazy_weak", "lazy_value", "reduce", "sqr_lazy", "mul_addhi", "sqr_addhi"):
        got = _ints(gpu.selftest_field(field, op, a, b))
        for (x, y), g in zip(pairs, got):
            assert g == _expected(op, x, y, p), (field, op, hex(x), hex(y))
    z = gpu.selftest_field(field, "zero_tests", a, b)
    for (x, y), r in zip(pairs, z):
        assert (int(r[0]) & 1) == int((x - y) % p == 0), (hex(x), hex(y))
        assert (int(r[0]) >> 1 & 1) == int((x - y) * x % p == 0), (hex(x), hex(y))


# ---------------------------------------------------------------------------------------------------------------------
def _norm_xyzz(oracle, r):
    """device result {X, Y, ZZ, ZZZ} -> normalised reference element (12 limbs)"""
    out = np.zeros(12, dtype=np.uint64)
    if not r[8:12].any():
        out[7] = np.uint64(1 << 63)
        return out
    out[0:4] = oracle.mul(FQ, r[0:4], oracle.invert(FQ, r[8:12]))
    out[4:8] = oracle.mul(FQ, r[4:8], oracle.invert(FQ, r[12:16]))
    out[8:12] = oracle.const(FQ, "one")
    # ZZ^3 == ZZZ^2: the pair is a consistent extended-Jacobian denominator
    zz, zzz = r[8:12], r[12:16]
    assert np.array_equal(oracle.mul(FQ, oracle.sqr(FQ, zz), zz), oracle.sqr(FQ, zzz))
    return out


def _norm(oracle, p):
    return oracle.g1_normalize_or_inf(p) if hasattr(oracle, "g1_normalize_or_inf") else oracle.g1_normalize(p)


def _inf():
    p = np.zeros(12, dtype=np.uint64)
    p[7] = np.uint64(1 << 63)
    return p


def test_device_g1_ops_vs_reference_outputs(gpu, oracle, golden):
    """g1_ops.json: mixed_add, add, dbl outputs of the reference's group.hpp on the same operands (compared after normalisation:
    the device uses other coordinates, the affine point is unique)"""
    cases = golden("g1_ops.json")["cases"]
    acc = np.stack([limbs(c["acc"]) for c in cases])
    q = np.stack([limbs(c["scalar_mul_G"]) for c in cases])
    m = np.stack([limbs(c["mixed_add"]) for c in cases])
    a = np.stack([limbs(c["add"]) for c in cases])
    for op, p_in, q_in, key in (("madd", acc, q, "mixed_add"), ("add", m, acc, "add"), ("quad_add", m, acc, "add"), ("dbl", a, a, "dbl")):
        got = gpu.selftest_g1(op, p_in, q_in)
        for c, r in zip(cases, got):
            assert np.array_equal(_norm_xyzz(oracle, r), oracle.g1_normalize(limbs(c[key]))), (op, c["scalar"])
    for c, r in zip(cases, gpu.selftest_g1("dbl", a, a)):
        assert np.array_equal(_norm_xyzz(oracle, r), limbs(c["normalize"]))


def test_device_g1_reference_kats(gpu, oracle, golden):
    """test_g1.cpp:41-122: mixed_add, add and three doublings on the hard-coded points"""
    def mont(d, keys):
        return np.concatenate([oracle.to_mont(FQ, limbs(d[c])) for c in keys])
    done = 0
    for k in golden("reference_kats.json")["g1"]:
        if k["op"] == "mixed_add":
            qa = np.concatenate([mont(k["b"], "xy"), np.zeros(4, dtype=np.uint64)])
            got = gpu.selftest_g1("madd", mont(k["a"], "xyz"), qa)[0]
        elif k["op"] == "add":
            got = gpu.selftest_g1("add", mont(k["a"], "xyz"), mont(k["b"], "xyz"))[0]
        elif k["op"] == "dbl3":
            cur = mont(k["a"], "xyz")
            for _ in range(3):
                cur = _norm_xyzz(oracle, gpu.selftest_g1("dbl", cur, cur)[0])
            assert np.array_equal(cur, oracle.g1_normalize(mont(k["expected"], "xyz"))), k["cite"]
            done += 1
            continue
        else:
            continue
        assert np.array_equal(_norm_xyzz(oracle, got), oracle.g1_normalize(mont(k["expected"], "xyz"))), k["cite"]
        done += 1
    assert done >= 3


def test_device_g1_exceptional_cases(gpu, oracle, golden):
    """test_g1.cpp:124-241 on the device layer: P + P, P + (-P), infinity operands, conditional negation"""
    cases = golden("g1_ops.json")["cases"][:6]
    jac = [limbs(c["dbl"]) for c in cases]                       # non-normalised Jacobian representatives
    aff = [oracle.g1_normalize(j) for j in jac]                  # the same points, z = one
    neg = []
    for a in aff:
        n = a.copy()
        n[4:8] = oracle.neg(FQ, a[4:8])
        neg.append(n)
    P, A, N = np.stack(jac), np.stack(aff), np.stack(neg)
    want_dbl = [oracle.g1_normalize(oracle.g1_dbl(j)) for j in jac]
    # mixed addition: P + P -> doubling branch; P + (-P) -> infinity (both via a negated y and via the negating entry); inf + Q -> Q
    for r, w in zip(gpu.selftest_g1("madd", P, A), want_dbl):
        assert np.array_equal(_norm_xyzz(oracle, r), w)
    for r in gpu.selftest_g1("madd", P, N):
        assert np.array_equal(_norm_xyzz(oracle, r), _inf())
    for r in gpu.selftest_g1("madd_neg", P, A):
        assert np.array_equal(_norm_xyzz(oracle, r), _inf())
    for r, w in zip(gpu.selftest_g1("madd_neg", P, N), want_dbl):  # P - (-P)
        assert np.array_equal(_norm_xyzz(oracle, r), w)
    INF = np.stack([_inf()] * len(jac))
    for r, a in zip(gpu.selftest_g1("madd", INF, A), aff):
        assert np.array_equal(_norm_xyzz(oracle, r), a)
    # mixed addition of two different points, negated: P_i - Q_{i+1}
    Q = np.roll(A, 1, axis=0)
    for r, j, n in zip(gpu.selftest_g1("madd_neg", P, Q), jac, np.roll(N, 1, axis=0)):
        assert np.array_equal(_norm_xyzz(oracle, r), oracle.g1_normalize(oracle.g1_mixed_add(j, n[:8])))
    # full addition: P + P, P + (-P), inf + P, P + inf, inf + inf -- by the one-lane add() and by the four-lane quad addition of the bucket reduction
    for op in ("add", "quad_add"):
        for r, w in zip(gpu.selftest_g1(op, P, A), want_dbl):
            assert np.array_equal(_norm_xyzz(oracle, r), w), op
        for r in gpu.selftest_g1(op, P, N):
            assert np.array_equal(_norm_xyzz(oracle, r), _inf()), op
        for r, a in zip(gpu.selftest_g1(op, INF, P), aff):
            assert np.array_equal(_norm_xyzz(oracle, r), a), op
        for r, a in zip(gpu.selftest_g1(op, P, INF), aff):
            assert np.array_equal(_norm_xyzz(oracle, r), a), op
        for r in gpu.selftest_g1(op, INF, INF):
            assert np.array_equal(_norm_xyzz(oracle, r), _inf()), op
        # two different points, in both orders, mixed with exceptional quads in the same wave
        Qr = np.roll(P, 1, axis=0)
        for r, j, k in zip(gpu.selftest_g1(op, P, Qr), jac, np.roll(np.stack(jac), 1, axis=0)):
            assert np.array_equal(_norm_xyzz(oracle, r), oracle.g1_normalize(oracle.g1_add(j, k))), op
    for r, w in zip(gpu.selftest_g1("add", P, A), want_dbl):
        assert np.array_equal(_norm_xyzz(oracle, r), w)
    for r in gpu.selftest_g1("add", P, N):
        assert np.array_equal(_norm_xyzz(oracle, r), _inf())
    for r, a in zip(gpu.selftest_g1("add", INF, P), aff):
        assert np.array_equal(_norm_xyzz(oracle, r), a)
    for r, a in zip(gpu.selftest_g1("add", P, INF), aff):
        assert np.array_equal(_norm_xyzz(oracle, r), a)
    for r in gpu.selftest_g1("add", INF, INF):
        assert np.array_equal(_norm_xyzz(oracle, r), _inf())
    # doubling: infinity stays infinity; the affine doubling equals the general one
    for r in gpu.selftest_g1("dbl", INF, INF):
        assert np.array_equal(_norm_xyzz(oracle, r), _inf())
    for r, a in zip(gpu.selftest_g1("dbl_affine", A, A), aff):
        assert np.array_equal(_norm_xyzz(oracle, r), oracle.g1_normalize(oracle.g1_dbl(a)))
