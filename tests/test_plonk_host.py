"""CPU tests of the resident PLONK prover's host side (no GPU): the StandardComposer mirror against the reference
composer's output, and the Fiat-Shamir transcript (Keccak-256 + host fr/fq arithmetic of libbbgpu.so) against the
challenges the reference's prover derived for the golden proofs."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from barretenberg_amd import BbGpu
from barretenberg_amd.plonk import bench_circuit, proof_from_lines, proof_lines

MAPS = ("sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


MIMC_X0 = 0x0777777788888888555555556666666633333333444444441111111122222222
MIMC_K = 0x0ABCDEFABCDEFABC1234123412341234DDDDEEEEFFFF00009999AAAABBBBCCCC


@pytest.mark.parametrize("gates", [3, 6, 30, 93, 4094])
def test_mimc_composer_mirror_digests(trace, gates):
    """MiMCComposer mirror (mimc_composer.cpp:13-250: chained MiMC gates, the no-op gates that carry a pending output wire, the closing
    gate) vs the reference composer's Prover state, SHA-256 of all thirteen arrays"""
    from barretenberg_amd.plonk import mimc_circuit
    st = mimc_circuit(gates, MIMC_X0, MIMC_K).preprocess()
    want = trace["mimc"]["input_digests"][str(gates)]
    assert st["n"] == want["n"]
    for k, d in want.items():
        if k != "n":
            assert hashlib.sha256(np.ascontiguousarray(st[k]).tobytes()).hexdigest() == d, k


def test_mimc_composer_noop_and_closing_gates():
    """a MiMC gate whose input is not the pending output wire gets a no-op gate in front of it, and a chain left open at preprocess() is
    closed by a gate constraining only w_o (mimc_composer.cpp:65-120, 172-190)"""
    from barretenberg_amd.plonk import MiMCComposer
    c = MiMCComposer()
    k, a, a3, a7, b, b3, b7 = (c.add_variable(v) for v in (5, 11, 12, 13, 21, 22, 23))
    c.create_mimc_gate(a, a3, k, a7, 9)
    c.create_mimc_gate(b, b3, k, b7, 10)  # b is not a7: no-op gate carrying a7 goes in between
    assert c.n == 3 and c.w_o == [a, a7, b] and c.q_mimc_selector == [1, 0, 1]
    st = c.preprocess()  # closing gate for b7
    assert st["n"] == 4 and c.w_o[3] == b7 and c.w_l[3] == c.zero_idx


@pytest.mark.parametrize("gates", [8, 32, 100, 160])
def test_extended_composer_state_fixture_is_a_satisfied_circuit(gates):
    """tests/golden/plonk_extended_state.npz (the reference ExtendedComposer's Prover input state) read with the plain numpy loader: every
    row satisfies the extended arithmetic identity q_m w_l w_r + q_l w_l + q_r w_r + q_o w_o + q_c + q_o_next w_o[i+1] = 0
    (arithmetic_widget.cpp:66-104 + sequential_widget.cpp:47-62) and the boolean constraints, and the sigma mappings are permutations"""
    from oracle.pyoracle import FR_MODULUS
    z = np.load(os.path.join(ROOT, "tests", "golden", "plonk_extended_state.npz"))
    st = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("%d/" % gates)}
    n = int(st["n"][0])
    rinv = pow(1 << 256, -1, FR_MODULUS)

    def vals(k):
        return [sum(int(v) << (64 * j) for j, v in enumerate(row)) * rinv % FR_MODULUS for row in st[k]]
    wl, wr, wo = vals("w_l"), vals("w_r"), vals("w_o")
    q = {k: vals(k) for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_bl", "q_br", "q_bo", "q_o_next")}
    assert any(q["q_o_next"]) and any(q["q_br"])
    for i in range(n):
        lhs = q["q_m"][i] * wl[i] * wr[i] + q["q_l"][i] * wl[i] + q["q_r"][i] * wr[i] + q["q_o"][i] * wo[i] + q["q_c"][i] + q["q_o_next"][i] * wo[(i + 1) % n]
        assert lhs % FR_MODULUS == 0, i
        for sel, w in ((q["q_bl"], wl), (q["q_br"], wr), (q["q_bo"], wo)):
            assert sel[i] * (w[i] * w[i] - w[i]) % FR_MODULUS == 0, i
    ids = sorted(int(v) for k in ("sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping") for v in st[k])
    assert ids == sorted(i + (t << 30) for t in range(3) for i in range(n))


def test_reference_bytes_of_an_infinity_commitment_are_only_pinned_in_the_flag(golden):
    """tests/golden/infinity_commitments.json (reference outputs, tools/gen_infinity_golden.py): for a commitment that is the point at
    infinity the reference sets the flag -- bit 63 of y.data[3] (group.hpp:133-151, re-set by normalize(), :450-468) -- and leaves the rest
    of the pair to whatever its accumulators held: the bytes change with the OpenMP thread count, in the verification key (Q_C of the
    bench circuit) and in a proof (W_R, W_O of the zero-wire circuit), where they also change the Fiat-Shamir challenge beta and with
    it Z_1.  Every such run verifies.  So the flag is the only thing there is to reproduce; the GPU half (what this library returns,
    and that the reference's Verifier accepts it) is tests/test_gpu_plonk.py::test_commitments_at_infinity."""
    fx = golden("infinity_commitments.json")
    threads = ("1", "4", "8")
    vk, pr = fx["verification_key_bench_circuit"], fx["proof_zerowire_circuit"]
    for t in threads:
        assert int(vk[t]["Q_C.y"][:16], 16) >> 63 == 1
        assert int(pr[t]["W_R.y"][:16], 16) >> 63 == 1 and int(pr[t]["W_O.y"][:16], 16) >> 63 == 1
        assert pr[t]["verified"] == "1"
        # finite commitments do not depend on the thread count
        assert vk[t]["Q_M.x"] == vk["1"]["Q_M.x"] and vk[t]["Q_M.y"] == vk["1"]["Q_M.y"]
        assert pr[t]["W_L.x"] == pr["1"]["W_L.x"] and pr[t]["W_L.y"] == pr["1"]["W_L.y"]
    # the bytes under the flag do
    assert len({vk[t]["Q_C.x"] for t in threads}) > 1 and len({vk[t]["Q_C.y"] for t in threads}) > 1
    assert len({pr[t]["W_R.x"] for t in threads}) > 1
    # ... and they reach the transcript: beta and the commitment computed from it differ between thread counts
    assert len({pr[t]["beta"] for t in threads}) > 1 and len({pr[t]["Z_1.x"] for t in threads}) > 1
    # the mirror of the zero-wire circuit has two all-zero wires
    from barretenberg_amd.plonk import zero_wire_circuit
    st = zero_wire_circuit(32, 0x0777777788888888555555556666666633333333444444441111111122222222).preprocess()
    assert st["n"] == fx["n"] and not np.any(st["w_r"]) and not np.any(st["w_o"]) and np.any(st["w_l"])
