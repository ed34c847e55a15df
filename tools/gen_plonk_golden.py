#!/usr/bin/env python3
"""Fixtures for the resident PLONK prover, produced by the reference itself (oracle/_ref/plonk_cpu = the reference's
unmodified composer / prover / verifier compiled in place by oracle/Makefile; driver oracle/plonk_driver.cpp):

  tests/golden/plonk_trace.json    Fiat-Shamir challenges (beta, gamma, alpha, z, nu) of the golden proofs, and SHA-256
                                   digests of the waffle::Prover input state (`plonk_cpu dump`) per circuit size, plus the
                                   complete input state of the 32-gate circuit
Run in the build container (needs /root/reference at oracle build time):  python tools/gen_plonk_golden.py
Circuits above 2^16 gates need a longer SRS than the 65,536-point oracle/_ref/transcript.dat: write one with
`cd $DIR && mkdir -p oracle/_ref && plonk_cpu transcript oracle/_ref/transcript.dat 1048576` (1 min) and pass BBGPU_BIG_SRS_DIR=$DIR;
the reference binary reads its transcript relative to the working directory.  Their proofs are appended to
tests/golden/plonk_proofs.json as well (the 32..2^16 entries of that file come from `plonk_cpu prove`).
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "oracle", "_ref", "plonk_cpu")
FIELDS64 = ("w_l", "w_r", "w_o")
MAPS = ("sigma_1_mapping", "sigma_2_mapping", "sigma_3_mapping")
SELECTORS = ("q_m", "q_l", "q_r", "q_o", "q_c")


def load_dump(path):
    b = open(path, "rb").read()
    assert b[:8] == b"BBPLONK1"
    n = int.from_bytes(b[8:16], "little")
    o, st = 16, {"n": n}
    for k in FIELDS64:
        st[k] = np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=o).reshape(n, 4); o += 32 * n
    for k in MAPS:
        st[k] = np.frombuffer(b, dtype=np.uint32, count=n, offset=o); o += 4 * n
    for k in SELECTORS:
        st[k] = np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=o).reshape(n, 4); o += 32 * n
    return st


def main():
    out = {"source": "oracle/_ref/plonk_cpu {trace,dump} <gates> (the reference's own composer and prover; see tools/gen_plonk_golden.py)",
           "witness_a0": "0777777788888888555555556666666633333333444444441111111122222222",
           "witness_b0": "0abcdefabcdefabc1234123412341234ddddeeeeffff00009999aaaabbbbcccc",
           "challenges": {}, "input_digests": {}, "verification_keys": {}}
    big = os.environ.get("BBGPU_BIG_SRS_DIR")
    proofs_path = os.path.join(ROOT, "tests", "golden", "plonk_proofs.json")
    proofs = json.load(open(proofs_path))
    for gates in (32, 1024, 16384, 65536) + ((262144, 1048576, 2097152) if big else ()):  # 2^21 gates: 4n = 2^23 transforms (three-pass NTT), needs a 2^21-point transcript
        r = subprocess.run([EXE, "trace", str(gates)], cwd=big if gates > 65536 else ROOT, capture_output=True, text=True, check=True)
        lines = r.stdout.strip().split("\n")
        if gates > 65536:
            proofs["proofs"][str(gates)] = [ln for ln in lines if ln.split()[0] not in ("beta", "gamma", "alpha", "z", "nu")]
        ch = {ln.split()[0]: ln.split()[1] for ln in lines if ln.split()[0] in ("beta", "gamma", "alpha", "z", "nu")}
        out["challenges"][str(gates)] = ch
        if gates <= 65536:  # waffle::preprocess(prover): SIGMA_1..3 and the widget's selector commitments (`plonk_cpu vk`)
            out["verification_keys"][str(gates)] = subprocess.run([EXE, "vk", str(gates)], cwd=ROOT, capture_output=True, text=True,
                                                                  check=True).stdout.strip().split("\n")
        path = "/tmp/plonk_dump_%d.bin" % gates
        subprocess.run([EXE, "dump", str(gates), path], cwd=big if gates > 65536 else ROOT, check=True, stdout=subprocess.DEVNULL)
        st = load_dump(path)
        out["input_digests"][str(gates)] = {k: hashlib.sha256(np.ascontiguousarray(st[k]).tobytes()).hexdigest() for k in FIELDS64 + MAPS + SELECTORS}
        out["input_digests"][str(gates)]["n"] = st["n"]
        if gates == 32:
            out["input_state_32"] = {k: [["%016x" % int(v) for v in row] for row in st[k]] for k in FIELDS64 + SELECTORS}
            out["input_state_32"].update({k: [int(v) for v in st[k]] for k in MAPS})
        os.remove(path)
    # BoolComposer circuit (arithmetic + bool widget; `BB_CIRCUIT=bool plonk_cpu ...`): proofs, challenges, verification keys, input digests
    benv = dict(os.environ, BB_CIRCUIT="bool")
    bsel = SELECTORS + ("q_bl", "q_br", "q_bo")
    out["bool"] = {"proofs": {}, "challenges": {}, "verification_keys": {}, "input_digests": {}}
    for gates in (2, 6, 14, 64, 4096):  # n = 4, 8, 16 (the reference's own smallest proofs, test_verifier.cpp:105-122), 128, 8192
        lines = subprocess.run([EXE, "trace", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True, env=benv).stdout.strip().split("\n")
        out["bool"]["challenges"][str(gates)] = {ln.split()[0]: ln.split()[1] for ln in lines if ln.split()[0] in ("beta", "gamma", "alpha", "z", "nu")}
        out["bool"]["proofs"][str(gates)] = [ln for ln in lines if ln.split()[0] not in ("beta", "gamma", "alpha", "z", "nu")]
        out["bool"]["verification_keys"][str(gates)] = subprocess.run([EXE, "vk", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True,
                                                                      env=benv).stdout.strip().split("\n")
        path = "/tmp/plonk_dump_bool_%d.bin" % gates
        subprocess.run([EXE, "dump", str(gates), path], cwd=ROOT, check=True, stdout=subprocess.DEVNULL, env=benv)
        b = open(path, "rb").read()
        n = int.from_bytes(b[8:16], "little")
        o, dig = 16, {"n": n}
        for k in FIELDS64:
            dig[k] = hashlib.sha256(b[o:o + 32 * n]).hexdigest(); o += 32 * n
        for k in MAPS:
            dig[k] = hashlib.sha256(b[o:o + 4 * n]).hexdigest(); o += 4 * n
        for k in bsel:
            dig[k] = hashlib.sha256(b[o:o + 32 * n]).hexdigest(); o += 32 * n
        out["bool"]["input_digests"][str(gates)] = dig
        os.remove(path)
    # MiMCComposer circuit (arithmetic + MiMC widget; `BB_CIRCUIT=mimc plonk_cpu ...`): a chain of MiMC rounds, then one addition gate
    menv = dict(os.environ, BB_CIRCUIT="mimc")
    msel = SELECTORS + ("q_mimc_selector", "q_mimc_coefficient")
    out["mimc"] = {"proofs": {}, "challenges": {}, "verification_keys": {}, "input_digests": {}}
    for gates in (3, 6, 30, 93, 4094):  # n = 4, 8, 32, 128 (91 rounds + noop + add gate pad to 128), 4096
        lines = subprocess.run([EXE, "trace", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True, env=menv).stdout.strip().split("\n")
        out["mimc"]["challenges"][str(gates)] = {ln.split()[0]: ln.split()[1] for ln in lines if ln.split()[0] in ("beta", "gamma", "alpha", "z", "nu")}
        out["mimc"]["proofs"][str(gates)] = [ln for ln in lines if ln.split()[0] not in ("beta", "gamma", "alpha", "z", "nu")]
        out["mimc"]["verification_keys"][str(gates)] = subprocess.run([EXE, "vk", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True,
                                                                      env=menv).stdout.strip().split("\n")
        path = "/tmp/plonk_dump_mimc_%d.bin" % gates
        subprocess.run([EXE, "dump", str(gates), path], cwd=ROOT, check=True, stdout=subprocess.DEVNULL, env=menv)
        b = open(path, "rb").read()
        n = int.from_bytes(b[8:16], "little")
        o, dig = 16, {"n": n}
        for k in FIELDS64:
            dig[k] = hashlib.sha256(b[o:o + 32 * n]).hexdigest(); o += 32 * n
        for k in MAPS:
            dig[k] = hashlib.sha256(b[o:o + 4 * n]).hexdigest(); o += 4 * n
        for k in msel:
            dig[k] = hashlib.sha256(b[o:o + 32 * n]).hexdigest(); o += 32 * n
        out["mimc"]["input_digests"][str(gates)] = dig
        os.remove(path)
    # ExtendedComposer circuit (arithmetic + sequential + bool widgets; `BB_CIRCUIT=extended plonk_cpu ...`).  The 700-line composer (gate
    # folding) is not mirrored in Python: the waffle::Prover INPUT state it produces (`plonk_cpu dump`) is stored as data in
    # tests/golden/plonk_extended_state.npz and handed to the resident prover as it is.
    xenv = dict(os.environ, BB_CIRCUIT="extended")
    xsel = SELECTORS + ("q_bl", "q_br", "q_bo", "q_o_next")
    out["extended"] = {"proofs": {}, "challenges": {}, "verification_keys": {}}
    xstate = {}
    for gates in (8, 32, 100, 160):  # n = 8, 32, 64, 128
        lines = subprocess.run([EXE, "trace", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True, env=xenv).stdout.strip().split("\n")
        out["extended"]["challenges"][str(gates)] = {ln.split()[0]: ln.split()[1] for ln in lines if ln.split()[0] in ("beta", "gamma", "alpha", "z", "nu")}
        out["extended"]["proofs"][str(gates)] = [ln for ln in lines if ln.split()[0] not in ("beta", "gamma", "alpha", "z", "nu")]
        out["extended"]["verification_keys"][str(gates)] = subprocess.run([EXE, "vk", str(gates)], cwd=ROOT, capture_output=True, text=True, check=True,
                                                                          env=xenv).stdout.strip().split("\n")
        path = "/tmp/plonk_dump_ext_%d.bin" % gates
        subprocess.run([EXE, "dump", str(gates), path], cwd=ROOT, check=True, stdout=subprocess.DEVNULL, env=xenv)
        b = open(path, "rb").read()
        n = int.from_bytes(b[8:16], "little")
        o = 16
        xstate["%d/n" % gates] = np.array([n], dtype=np.uint64)
        for k in FIELDS64:
            xstate["%d/%s" % (gates, k)] = np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=o).reshape(n, 4).copy(); o += 32 * n
        for k in MAPS:
            xstate["%d/%s" % (gates, k)] = np.frombuffer(b, dtype=np.uint32, count=n, offset=o).copy(); o += 4 * n
        for k in xsel:
            xstate["%d/%s" % (gates, k)] = np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=o).reshape(n, 4).copy(); o += 32 * n
        assert o == len(b)
        os.remove(path)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "plonk_extended_state.npz"), **xstate)
    if not big:  # keep the large-circuit entries of an earlier run
        old = json.load(open(os.path.join(ROOT, "tests", "golden", "plonk_trace.json")))
        for key in ("challenges", "input_digests", "verification_keys"):
            for g, v in old.get(key, {}).items():
                out[key].setdefault(g, v)
    else:
        with open(proofs_path, "w") as fh:
            json.dump(proofs, fh, indent=0)
    with open(os.path.join(ROOT, "tests", "golden", "plonk_trace.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote tests/golden/plonk_trace.json")


if __name__ == "__main__":
    main()
