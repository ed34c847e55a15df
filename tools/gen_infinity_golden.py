#!/usr/bin/env python3
"""tests/golden/infinity_commitments.json: what the REFERENCE emits for a commitment that is the point at infinity.
Runs the all-CPU reference build (oracle/_ref/plonk_cpu, compiled from /root/reference by oracle/Makefile) in the build container:
  * the verification key of the bench circuit (its q_c selector is identically zero -> Q_C = infinity), 1 / 4 / 8 OpenMP threads;
  * a proof of the `zerowire` circuit of oracle/plonk_driver.cpp (w_r = w_o = 0 -> W_R = W_O = infinity), 1 / 4 / 8 threads.
Finding recorded by the fixture: bit 63 of y.data[3] (the infinity flag, group.hpp:133-151) is always set -- normalize() re-sets it
(group.hpp:450-468) -- and every OTHER bit of the pair is whatever the accumulators of that run held: it changes with the thread
count, and since the proof bytes enter the Fiat-Shamir hash so do beta and everything after it.  The reference therefore has no
reproducible bytes for such a proof; each run is self-consistent and verifies."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "plonk_cpu")


def run(args, threads, circuit=None):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    if circuit:
        env["BB_CIRCUIT"] = circuit
    r = subprocess.run([EXE] + args, cwd=ROOT, capture_output=True, text=True, env=env, check=False)
    return dict(ln.split() for ln in r.stdout.strip().split("\n") if len(ln.split()) == 2)


out = {"generator": "tools/gen_infinity_golden.py (reference build oracle/_ref/plonk_cpu)", "gates": 32, "verification_key_bench_circuit": {}, "proof_zerowire_circuit": {}}
for t in (1, 4, 8):
    vk = run(["vk", "32"], t)
    out["verification_key_bench_circuit"][str(t)] = {k: vk[k] for k in ("Q_C.x", "Q_C.y", "Q_M.x", "Q_M.y")}
    pr = run(["trace", "32"], t, "zerowire")
    out["n"] = int(pr["n"])  # 32 gates of the zero-wire circuit -> n = 64
    out["proof_zerowire_circuit"][str(t)] = {k: pr[k] for k in ("W_L.x", "W_L.y", "W_R.x", "W_R.y", "W_O.x", "W_O.y", "Z_1.x", "beta", "verified")}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "infinity_commitments.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
