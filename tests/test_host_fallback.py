"""The error contract of the drop-in boundary (SURVEY 8b "C ABI underneath", "Errors"): the reference's signatures cannot report a failure
(assert.hpp:13-23, scalar_multiplication.cpp:680-684), so when a GPU call fails at run time the C++ shim computes the result with the library's
own host code (csrc/host_fallback.hpp through the bbgpu_host_* entries) and carries on.  CPU tests -- this container has no GPU, which IS the
failure: (i) the host entries against the reference's fixtures and the oracle, (ii) the reference's unmodified prover linked on the shim
proves and verifies without a device, bit-identical to the all-CPU reference build, (iii) BBGPU_SHIM_STRICT=1 aborts instead.
(The GPU tests run the shim with BBGPU_SHIM_STRICT=1, so a host answer can never stand in for a kernel there.)"""
import os
import subprocess

import numpy as np
import pytest

from oracle.pyoracle import FR_MODULUS, NTT_KINDS, PolyOracle as P, aligned_copy, to_int
from tests.util import CONST_SEED, NTT_SEED, SCALAR_SEED, limbs, noncanonical, sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from barretenberg_amd import BbGpu
    return BbGpu(init=False)  # never binds a device


def hx(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return ["%064x" % to_int(r) for r in a]


@pytest.fixture(scope="module")
def srs(oracle, golden):
    g = golden("msm.json")
    n = 10000
    pts = oracle.make_srs(limbs(g["srs_secret_mont"]), n)
    return g, pts, oracle.point_table(pts), oracle.random_scalars(SCALAR_SEED, n)


def test_host_msm_reference_fixtures(lib, srs):
    """the reference's own pippenger() results (tests/golden/msm.json) for every plain case up to 10,000 points"""
    g, pts, table, scalars = srs
    done = 0
    for case in g["cases"]:
        n = case["n"]
        if n > 10000 or "scalars" in case or "points" in case or "infinity" in case:
            continue
        out = lib.host_msm(scalars, table, n)
        assert np.array_equal(out[0:4], limbs(case["x"])) and np.array_equal(out[4:8], limbs(case["y"])), n
        assert not (int(out[7]) >> 63)
        done += 1
    assert done >= 9
    assert int(lib.host_msm(scalars, table, 0)[7]) >> 63 == 1
    assert int(lib.host_msm(aligned_copy(np.zeros((100, 4), dtype=np.uint64)), table, 100)[7]) >> 63 == 1


@pytest.mark.parametrize("n", [33, 64, 257, 1029])
def test_host_msm_vs_oracle(lib, oracle, srs, n):
    """ragged sizes, [r, 2r) representatives, sub-slices as the reference passes them, the plain n-entry table, P - P"""
    g, pts, table, scalars = srs
    want = oracle.msm_affine(scalars, table, n)
    assert np.array_equal(lib.host_msm(scalars, table, n)[:8], want[:8])
    assert np.array_equal(lib.host_msm(aligned_copy(noncanonical(scalars[:n], FR_MODULUS)), table, n)[:8], want[:8])
    assert np.array_equal(lib.host_msm(scalars, aligned_copy(pts[:n]), n, plain=True)[:8], want[:8])
    want = oracle.msm_affine(aligned_copy(scalars[7:n]), aligned_copy(table[14:2 * n]), n - 7)
    assert np.array_equal(lib.host_msm(aligned_copy(scalars[7:n]), table[14:], n - 7)[:8], want[:8])
    # every point equal, scalars +1 / -1 alternating: the sum cancels through the exceptional branches
    from oracle.pyoracle import FR
    one = oracle.const(FR, "one")
    same_t = oracle.point_table(aligned_copy(np.tile(pts[3], (64, 1))))
    sc = aligned_copy(np.stack([one, oracle.neg(FR, one)] * 32))
    assert int(lib.host_msm(sc, same_t, 64)[7]) >> 63 == 1


def test_host_ntt_reference_fixtures(lib, oracle, golden):
    """outputs of the reference itself: n = 2 .. 16 in full, SHA-256 digests + samples at 2^8 .. 2^16, all seven entry points"""
    g = golden("ntt.json")
    c = limbs(g["constant"])
    for case in g["small"]:
        co = limbs(case["input"]).reshape(-1, 4)
        assert np.array_equal(lib.host_ntt(co.copy(), case["kind"], c).reshape(-1), limbs(case["output"])), (case["n"], case["kind"])
    done = 0
    for n in (256, 1024, 65536):
        co = noncanonical(oracle.random_scalars(NTT_SEED, n), FR_MODULUS)
        for case in [x for x in g["large"] if x["n"] == n]:
            got = lib.host_ntt(co.copy(), case["kind"], c)
            for i, v in case["samples"].items():
                assert np.array_equal(got[int(i)], limbs(v)), (n, case["kind"], i)
            assert sha(got) == case["sha256"], (n, case["kind"])
            done += 1
    assert done >= 21


@pytest.mark.parametrize("log2n", [1, 5, 9])
def test_host_ntt_vs_oracle_all_kinds(lib, oracle, log2n):
    n = 1 << log2n
    const = oracle.random_scalars(CONST_SEED, 1)[0]
    co = noncanonical(oracle.random_scalars(NTT_SEED + log2n, n), FR_MODULUS)
    for kind in NTT_KINDS:
        assert np.array_equal(lib.host_ntt(co.copy(), kind, const), oracle.ntt(co, kind, const)), (log2n, kind)


def test_host_poly_helpers_match_reference_fixtures(lib, oracle, golden):
    """evaluate / compute_kate_opening_coefficients / compute_lagrange_polynomial_fft / divide_by_pseudo_vanishing_polynomial against the
    outputs of the reference's own functions (tests/golden/poly_ops.json)"""
    for case in golden("poly_ops.json")["cases"]:
        n, seed = case["n"], case["seed"]
        v, z = oracle.random_scalars(seed, n), oracle.random_scalars(seed + 1, 1)[0]
        assert hx(lib.host_evaluate(v, z))[0] == case["evaluate"]
        dest, f = lib.host_kate_opening(v, z)
        assert hx(dest) == case["kate_dest"] and hx(f)[0] == case["kate_f"]
        assert hx(lib.host_lagrange_l1_fft(n, 2 * n)) == case["lagrange_l1_fft_2n"]
        for k, off, key in ((2, 3, "divide_vanishing_2n"), (4, 4, "divide_vanishing_4n")):
            c = oracle.random_scalars(seed + off, k * n)
            assert hx(lib.host_divide_by_pseudo_vanishing(c.copy(), n, k * n)) == case[key]


@pytest.mark.parametrize("n", [1, 2, 7, 257, 2049])
def test_host_poly_helpers_vs_oracle(lib, oracle, n):
    v = noncanonical(oracle.random_scalars(0x5CA9 + n, n), FR_MODULUS)
    z = oracle.random_scalars(0x5CAA + n, 1)[0]
    assert np.array_equal(lib.host_evaluate(v, z), P.evaluate(v, z))
    dest, f = lib.host_kate_opening(v, z)
    want, wf = P.kate_opening(v, z)
    assert np.array_equal(dest, want) and np.array_equal(f, wf)
    m = 1 << max(1, n.bit_length() - 1)  # the domain helpers want powers of two
    for k in (1, 4):
        c = oracle.random_scalars(0xD1F + m + k, k * m)
        assert np.array_equal(lib.host_divide_by_pseudo_vanishing(c.copy(), m, k * m), P.divide_by_pseudo_vanishing(c, m, k * m)), (m, k)
        assert np.array_equal(lib.host_lagrange_l1_fft(m, k * m), P.lagrange_l1_fft(m, k * m)), (m, k)


def _prover(build, gates, **env):
    exe = os.path.join(ROOT, "oracle", "_ref", build)
    srs = os.path.join(ROOT, "oracle", "_ref", "transcript.dat")
    if not (os.path.exists(exe) and os.path.exists(srs)):
        pytest.skip("oracle/_ref/%s is built from /root/reference by __graft_entry__.build() (build container only)" % build)
    return subprocess.run([exe, "prove", str(gates)], cwd=ROOT, capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, OMP_NUM_THREADS="4", **env))


@pytest.mark.parametrize("build,gates", [("plonk_gpu_full", 32), ("plonk_gpu", 32), ("plonk_gpu_full", 1024)])
def test_reference_prover_on_the_shim_without_a_gpu(golden, build, gates):
    """the reference's UNMODIFIED composer -> Prover -> Verifier linked on libbbshim.so, on a machine whose GPU calls all fail: every hot-path call
    is answered by the library's host code, the proof equals the all-CPU reference build's byte for byte and verifies.  plonk_gpu_full has
    scalar_multiplication.o / polynomial_arithmetic.o dropped from the link, so evaluate / kate / L_1 / the vanishing division take the same road."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the calls succeed (tests/test_gpu_parity.py runs this link with BBGPU_SHIM_STRICT=1)")
    r = _prover(build, gates)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().split("\n") == golden("plonk_proofs.json")["proofs"][str(gates)]
    assert "computing on the host" in r.stderr and "no HIP device" in r.stderr  # said so, with the library's own error text


def test_strict_shim_aborts_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = _prover("plonk_gpu_full", 32, BBGPU_SHIM_STRICT="1")
    assert r.returncode != 0 and "failed (-1)" in r.stderr and "computing on the host" not in r.stderr
