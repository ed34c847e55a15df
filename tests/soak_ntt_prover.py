#!/usr/bin/env python3
"""Soak run (not collected by pytest; `python tests/soak_ntt_prover.py [seconds]` on the GPU box): (1) transforms of random sizes 2 .. 2^16 and random kinds
against the oracle, interleaved on two streams with transforms of other sizes in flight (the shared scratch / domain-table paths); (2) the resident prover
re-proving the same 2^12-gate circuit: every proof must equal the first byte for byte (the overlap of transforms with commitments must never change a result)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu
from oracle.pyoracle import NTT_KINDS, Oracle, build

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
build()
O = Oracle()
G = BbGpu(0)
G.set_host_thresholds(0, 0)
rng = np.random.default_rng(int(time.time()))
kinds = list(NTT_KINDS)
const = O.random_scalars(3, 1)[0]
pool = O.random_scalars(4, 1 << 16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = cases = 0
t_end = time.time() + budget * 0.6
t_say = time.time() + 60  # a progress line a minute: a GPU command silent for 7 minutes is taken to be hung and killed
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d transforms so far, %d mismatches" % (cases, bad), flush=True); t_say = time.time() + 60
    lg, lg2 = int(rng.integers(1, 17)), int(rng.integers(8, 17))
    n, n2 = 1 << lg, 1 << lg2
    kind, kind2 = kinds[int(rng.integers(0, len(kinds)))], kinds[int(rng.integers(0, len(kinds)))]
    a = pool[rng.permutation(1 << 16)[:n]].copy()
    b = pool[rng.permutation(1 << 16)[:n2]].copy()
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    torch.cuda.synchronize()
    G.ntt_device(db.data_ptr(), n2, kind2, const, stream=s2.cuda_stream)
    G.ntt_device(da.data_ptr(), n, kind, const, stream=s1.cuda_stream)
    G.ntt_device(db.data_ptr(), n2, "fft", stream=s2.cuda_stream)
    torch.cuda.synchronize()
    want = O.ntt(a.copy(), kind, const)
    want2 = O.ntt(O.ntt(b.copy(), kind2, const), "fft")
    cases += 2
    if not np.array_equal(da.cpu().numpy().view(np.uint64), want):
        bad += 1; print("MISMATCH ntt 2^%d %s" % (lg, kind), flush=True)
    if not np.array_equal(db.cpu().numpy().view(np.uint64), want2):
        bad += 1; print("MISMATCH ntt 2^%d %s then fft" % (lg2, kind2), flush=True)
print("transforms: %d against the oracle, %d mismatches" % (cases, bad), flush=True)

from barretenberg_amd.plonk import FR_MODULUS, Prover, bench_circuit, to_montgomery_limbs  # noqa: E402
state = bench_circuit(1 << 12, 3, 5).preprocess()
srs = G.srs_generate(to_montgomery_limbs([0x1234567890ABCDEF1234567890ABCDEF % FR_MODULUS])[0], state["n"])
P = Prover(G, state, srs)
first = P.construct_proof()
t_end = time.time() + budget * 0.4
proofs = diff = 0
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d proofs so far, %d differing" % (proofs, diff), flush=True); t_say = time.time() + 60
    p = P.construct_proof()
    proofs += 1
    if not np.array_equal(p, first):
        diff += 1
print("prover: %d proofs of the same 2^12-gate circuit, %d differing from the first" % (proofs, diff), flush=True)
bad += diff
sys.exit(1 if bad else 0)
