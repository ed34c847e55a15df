"""CPU tests of the resident PLONK prover's host side (no GPU): the StandardComposer mirror against the reference
composer's output, and the Fiat-Shamir transcript (Keccak-256 + host fr/fq arithmetic of libbbgpu.so) against the
challenges the reference's prover derived for the golden proofs."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from barretenberg_amd import BbGpu
from barretenberg_amd.plonk import bench_circuit, proof_from_lines, proof_lines

MAPS = ("sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping")


@pytest.fixture(scope="module")
def trace(golden):
    return golden("plonk_trace.json")


def witnesses(trace):
    return int(trace["witness_a0"], 16), int(trace["witness_b0"], 16)


def test_composer_mirror_full_state_32(trace):
    """every value of the waffle::Prover input state of the 32-gate bench circuit (standard_composer.cpp:163-220)"""
    st = bench_circuit(32, *witnesses(trace)).preprocess()
    want = trace["input_state_32"]
    assert st["n"] == 32
    for k, v in want.items():
        if k in MAPS:
            assert st[k].tolist() == v, k
        else:
            assert [["%016x" % int(x) for x in row] for row in st[k]] == v, k


@pytest.mark.parametrize("gates", [32, 1024, 16384, 65536, 262144])
def test_composer_mirror_digests(trace, gates):
    st = bench_circuit(gates, *witnesses(trace)).preprocess()
    want = trace["input_digests"][str(gates)]
    assert st["n"] == want["n"]
    for k, d in want.items():
        if k != "n":
            assert hashlib.sha256(np.ascontiguousarray(st[k]).tobytes()).hexdigest() == d, k


@pytest.mark.parametrize("gates", [32, 1024, 16384, 65536, 262144, 1048576])
def test_transcript_challenges_match_reference(golden, trace, gates):
    """challenge.hpp:64-112 restated in plonk.hip (host code): gamma, beta, alpha, z recomputed from the reference's golden proof"""
    lines = golden("plonk_proofs.json")["proofs"][str(gates)]
    n, proof = proof_from_lines(lines)
    assert proof_lines(n, proof) == lines[:26]
    G = BbGpu(init=False)
    out = np.zeros(16, dtype=np.uint64)
    fn = G.lib.bbgpu_plonk_challenges_from_proof
    fn.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    assert fn(proof.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    want = trace["challenges"][str(gates)]
    for i, name in enumerate(("gamma", "beta", "alpha", "z")):
        got = "%016x%016x%016x%016x" % tuple(int(v) for v in out[4 * i:4 * i + 4][::-1])
        assert got == want[name], name


@pytest.mark.parametrize("gates", [2, 6, 14, 64, 4096])
def test_bool_composer_mirror_digests(trace, gates):
    """BoolComposer mirror (bool_composer.cpp:68-143) vs the reference composer's Prover state, SHA-256 of all fourteen arrays"""
    from barretenberg_amd.plonk import bool_circuit
    st = bool_circuit(gates).preprocess()
    want = trace["bool"]["input_digests"][str(gates)]
    assert st["n"] == want["n"]
    for k, d in want.items():
        if k != "n":
            assert hashlib.sha256(np.ascontiguousarray(st[k]).tobytes()).hexdigest() == d, k
