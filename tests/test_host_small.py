"""SURVEY 8b "small sizes" on the host (csrc/host_small.hpp): the Verifier's ~20-point MSM and the n <= 16 transforms of the
smallest circuits are answered without a device.  CPU tests: through the C ABI vs the oracle and the reference's fixtures.
(The same sizes on the GPU kernels: tests/test_gpu_parity.py, whose fixture switches the host path off.)"""
import numpy as np
import pytest

from oracle.pyoracle import FQ, FR, FR_MODULUS, NTT_KINDS, aligned_copy
from tests.util import CONST_SEED, NTT_SEED, SCALAR_SEED, limbs, noncanonical, sha


@pytest.fixture(scope="module")
def lib():
    from barretenberg_amd import BbGpu
    g = BbGpu(init=False)  # never binds a device: everything below must be answered on the host
    g.set_host_thresholds(32, 16)  # the host code is exercised up to 32 points here (the shipped default is 24: the measured crossover)
    return g


@pytest.fixture(scope="module")
def small_srs(oracle, golden):
    g = golden("msm.json")
    srs = oracle.make_srs(limbs(g["srs_secret_mont"]), 64)
    return g, srs, oracle.point_table(srs), oracle.random_scalars(SCALAR_SEED, 64)


def _check(out, case):
    if "infinity" in case:
        assert bool(int(out[7]) >> 63) == case["infinity"]
    else:
        assert np.array_equal(out[0:4], limbs(case["x"])) and np.array_equal(out[4:8], limbs(case["y"])), case
        assert not (int(out[7]) >> 63)


def test_host_msm_reference_fixtures(lib, oracle, small_srs):
    """the reference's own pippenger() results for n = 0, 1, 2, 3, 16 (tests/golden/msm.json) and its edge cases"""
    g, srs, table, scalars = small_srs
    one = oracle.const(FQ, "one")
    done = 0
    for case in g["cases"]:
        n = case["n"]
        if n > 32:
            continue
        if case.get("scalars") == "all zero":
            out = lib.pippenger(aligned_copy(np.zeros((16, 4), dtype=np.uint64)), table, 16)
        elif "points" in case:
            same_t = oracle.point_table(aligned_copy(np.tile(srs[5], (32, 1))))
            sc = aligned_copy(np.tile(oracle.const(FR, "one"), (32, 1))) if case.get("scalars") == "all one" else scalars
            out = lib.pippenger(sc, same_t, 32)
        elif "scalars" in case:
            continue
        else:
            out = lib.pippenger(scalars, table, n)
        _check(out, case)
        if "x" in case:
            assert np.array_equal(out[8:12], one)  # normalised: z = fq::one
        done += 1
    assert done >= 6


@pytest.mark.parametrize("n", [1, 2, 5, 20, 31, 32])
def test_host_msm_vs_oracle(lib, oracle, small_srs, n):
    g, srs, table, scalars = small_srs
    want = oracle.msm_affine(scalars, table, n)
    assert np.array_equal(lib.pippenger(scalars, table, n)[:8], want[:8])
    # scalars in [r, 2r) (polynomial_arithmetic.cpp:580-588) name the same point; sub-slices points + 2*off as the reference passes them
    assert np.array_equal(lib.pippenger(aligned_copy(noncanonical(scalars[:n], FR_MODULUS)), table, n)[:8], want[:8])
    if n >= 5:
        want = oracle.msm_affine(aligned_copy(scalars[3:n]), aligned_copy(table[6:2 * n]), n - 3)
        assert np.array_equal(lib.pippenger(aligned_copy(scalars[3:n]), table[6:], n - 3)[:8], want[:8])


def test_host_msm_batched_and_skew(lib, oracle, small_srs):
    g, srs, table, scalars = small_srs
    outs = lib.batched_scalar_multiplications([(table, aligned_copy(scalars[o:o + 20]), 20) for o in (0, 20, 40)])
    for o, out in zip((0, 20, 40), outs):
        assert np.array_equal(out[:8], oracle.msm_affine(aligned_copy(scalars[o:o + 20]), table, 20)[:8])
    one = oracle.const(FR, "one")
    sc = aligned_copy(np.stack([one, oracle.neg(FR, one)] * 8))  # P0 - P1 + P2 - ...
    assert np.array_equal(lib.pippenger(sc, table, 16)[:8], oracle.msm_affine(sc, table, 16)[:8])
    # P + (-P) = infinity through the exceptional branch of the mixed addition
    same_t = oracle.point_table(aligned_copy(np.tile(srs[3], (2, 1))))
    assert int(lib.pippenger(aligned_copy(sc[:2]), same_t, 2)[7]) >> 63 == 1


def test_host_ntt_reference_fixtures(lib, golden):
    """outputs of the reference itself for n = 2 .. 16, all seven entry points (tests/golden/ntt.json)"""
    g = golden("ntt.json")
    c = limbs(g["constant"])
    done = 0
    for case in g["small"]:
        if case["n"] > 16:
            continue
        co = limbs(case["input"]).reshape(-1, 4)
        got = lib.ntt(co.copy(), case["kind"], c)
        assert np.array_equal(got.reshape(-1), limbs(case["output"])), (case["n"], case["kind"])
        done += 1
    assert done >= 20


@pytest.mark.parametrize("log2n", [1, 2, 3, 4])
def test_host_ntt_vs_oracle_all_kinds(lib, oracle, log2n):
    n = 1 << log2n
    const = oracle.random_scalars(CONST_SEED, 1)[0]
    co = noncanonical(oracle.random_scalars(NTT_SEED + log2n, n), FR_MODULUS)
    for kind in NTT_KINDS:
        assert np.array_equal(lib.ntt(co.copy(), kind, const), oracle.ntt(co, kind, const)), (log2n, kind)


def test_nothing_larger_runs_on_the_host(lib):
    """above the thresholds the library needs its GPU and says so (no silent CPU path)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from barretenberg_amd import BbGpuError
    with pytest.raises(BbGpuError, match="no HIP device"):
        lib.fft(np.zeros((32, 4), dtype=np.uint64))
    with pytest.raises(BbGpuError, match="no HIP device"):
        lib.pippenger(np.zeros((33, 4), dtype=np.uint64), np.zeros((66, 8), dtype=np.uint64), 33)
    lib.set_host_thresholds(0, 0)
    try:
        with pytest.raises(BbGpuError, match="no HIP device"):
            lib.fft(np.zeros((4, 4), dtype=np.uint64))
        with pytest.raises(BbGpuError, match="no HIP device"):
            lib.pippenger(np.zeros((4, 4), dtype=np.uint64), np.zeros((8, 8), dtype=np.uint64), 4)
    finally:
        lib.set_host_thresholds(32, 16)


def test_host_code_under_sanitizers(oracle):
    """the host-side product code (host_small / host_g1 / host_g2 / host_fr / keccak) built with AddressSanitizer + UBSan (CPU build: the pool
    offers no GPU sanitizer) and driven against the oracle: msm_small at n = 0 .. 40 over four scalar mixtures incl. [r, 2r) representatives,
    batch normalisation, ntt_small for all seven kinds at 2 .. 64 elements, G2 additivity, Keccak-256"""
    import os
    import subprocess
    import tempfile
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tempfile.gettempdir(), "bbgpu_test_host_sanitize")
    ob = os.path.join(ROOT, "oracle", "_build")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
           os.path.join(ROOT, "tests", "cpp", "test_host_sanitize.cpp"), "-L" + ob, "-loracle", "-Wl,-rpath," + ob, "-pthread"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr or "ubsan" in b.stderr or "sanitize" in b.stderr):
        pytest.skip("no sanitizer runtime in this image: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
