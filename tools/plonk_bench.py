#!/usr/bin/env python3
"""BASELINE config 5 timing: construct_proof() of the resident prover (bbgpu_plonk_*) on the bench_plonk.cpp circuit,
next to the reference's own prover on the box's host cores (oracle/_ref/plonk_cpu, all-CPU) and the reference prover linked
against the shim (oracle/_ref/plonk_gpu) when those test-only binaries travelled with the repo.
    python tools/plonk_bench.py [--gates 65536] [--reps 10]"""
import argparse
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from barretenberg_amd import BbGpu  # noqa: E402
from barretenberg_amd.plonk import FR_MODULUS, Prover, bench_circuit, to_montgomery_limbs  # noqa: E402

A0 = 0x0777777788888888555555556666666633333333444444441111111122222222
B0 = 0x0ABCDEFABCDEFABC1234123412341234DDDDEEEEFFFF00009999AAAABBBBCCCC
SECRET = 0x0123456789ABCDEF0F1E2D3C4B5A6978FEDCBA98765432100123456789ABCDEF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gates", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-reference", action="store_true")
    args = ap.parse_args()
    G = BbGpu(0)
    t0 = time.perf_counter()
    state = bench_circuit(args.gates, A0, B0).preprocess()
    n = state["n"]
    print("circuit: %d gates -> n = %d (composer mirror %.2f s)" % (args.gates, n, time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    srs = G.srs_generate(to_montgomery_limbs([SECRET % FR_MODULUS])[0], n)
    print("SRS x^i G, %d points + window tables: %.1f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
    P = Prover(G, state, srs)
    t0 = time.perf_counter()
    first = P.construct_proof()
    print("first construct_proof (incl. circuit-only preparation %.2f ms): %.2f ms" % (P.timing()["first_use_preparation_ms"], (time.perf_counter() - t0) * 1e3))
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        p = P.construct_proof()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert np.array_equal(p, first)
    tm = P.timing()
    print("construct_proof, steady state: median %.2f ms  min %.2f ms  (last: commitments %.2f ms, transforms + pointwise + host %.2f ms)" % (
        float(np.median(ts)), min(ts), tm["commitments_ms"], tm["rest_ms"]), flush=True)
    if not args.no_reference and args.gates <= 65536:
        env = dict(os.environ, OMP_NUM_THREADS="16")
        for exe in ("plonk_cpu", "plonk_gpu"):
            path = os.path.join(ROOT, "oracle", "_ref", exe)
            if os.path.exists(path):
                best = None
                for _ in range(3):
                    r = subprocess.run([path, "prove", str(args.gates)], cwd=ROOT, capture_output=True, text=True, env=env)
                    m = re.search(r"construct_proof ([0-9.]+) ms", r.stderr)
                    if m:
                        best = float(m.group(1)) if best is None else min(best, float(m.group(1)))
                print("reference %s construct_proof (16 threads): %s ms" % (exe, best), flush=True)
    P.destroy()
    G.shutdown()


if __name__ == "__main__":
    main()
