#!/usr/bin/env python3
"""Soak run (not collected by pytest; `python tests/soak_msm.py [seconds]` on the GPU box): random sizes and scalar mixtures against the oracle for a
fixed wall-clock budget, plus pipelined batches whose results must equal the one-at-a-time results.  Looks for what a single pass of the parity
suite cannot: launch-to-launch nondeterminism (a missing barrier in the LDS-staged sort passes, a race between the MSM slots)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from barretenberg_amd import BbGpu, BbGpuError
from oracle.pyoracle import FR, Oracle, aligned_copy, build

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
build()
O = Oracle()
G = BbGpu(0)
G.set_host_thresholds(0, 0)
rng = np.random.default_rng(int(time.time()))
N = 1 << 15
x = O.random_scalars(11, 1)[0]
srs = O.make_srs(x, N)
table = O.point_table(srs)
pool = O.random_scalars(12, N)
one = O.const(FR, "one"); minus_one = O.neg(FR, one); zero = np.zeros(4, dtype=np.uint64)
t_end = time.time() + budget
cases = bad = 0
t_say = time.time() + 60  # a progress line a minute: a GPU command silent for 7 minutes is taken to be hung and killed
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d oracle comparisons so far, %d mismatches" % (cases, bad), flush=True); t_say = time.time() + 60
    n = int(rng.integers(25, 20000))
    if rng.integers(0, 3) == 0:
        n &= ~7
        n = max(n, 32)
    sc = pool[rng.permutation(N)[:n]].copy()
    kind = int(rng.integers(0, 5))
    if kind == 1:
        pick = rng.integers(0, 4, n); sc[pick == 0] = zero; sc[pick == 1] = one; sc[pick == 2] = minus_one
    elif kind == 2:
        sc[:] = pool[int(rng.integers(0, N))]
    elif kind == 3:
        sc = np.repeat(pool[: (n + 31) // 32], 32, axis=0)[:n].copy()
    sc = aligned_copy(sc)
    want = O.msm_affine(sc, table, n)
    got = G.pippenger(sc, table, n)
    cases += 1
    ok = (int(got[7]) >> 63) == (int(want[7]) >> 63) and ((int(want[7]) >> 63) or np.array_equal(got[:8], want[:8]))
    if not ok:
        bad += 1
        print("MISMATCH n=%d kind=%d" % (n, kind), flush=True)
print("oracle comparisons: %d cases, %d mismatches" % (cases, bad), flush=True)

# shares (round 3): N bucket-range shares, N row-range shares and N point-range shares of random sizes / scalar mixtures must add up to the one-call result
t2 = time.time() + budget / 4
share_cases = share_bad = 0
while time.time() < t2:
    if time.time() > t_say:
        print("... %d share splits so far, %d mismatches" % (share_cases, share_bad), flush=True); t_say = time.time() + 60
    n = int(rng.integers(1024, 20000))
    if rng.integers(0, 2) == 0:
        n &= ~7
    sc = pool[rng.permutation(N)[:n]].copy()
    kind = int(rng.integers(0, 4))
    if kind == 1:
        pick = rng.integers(0, 4, n); sc[pick == 0] = zero; sc[pick == 1] = one; sc[pick == 2] = minus_one
    elif kind == 2:
        sc[:] = pool[int(rng.integers(0, N))]
    tab = aligned_copy(table[:2 * n])
    # round 4: half of the tables are cut into several window-table segments (testing knob; the MSMs over them run as pieces on the ticket's slot
    # and a helper slot), the one-call result itself is then checked against the oracle
    segmented = rng.integers(0, 2) == 0
    if segmented:
        os.environ["BBGPU_TABLE_SEG_POINTS"] = str(int(rng.integers(max(64, n // 60), n)))
    h = G.srs_register(tab)
    os.environ.pop("BBGPU_TABLE_SEG_POINTS", None)
    d = torch.from_numpy(aligned_copy(sc).view(np.int64)).cuda()
    full = G.msm_device(h, d.data_ptr(), n)
    W = G.srs_num_windows(h, n)
    Nn = int(rng.integers(1, 9))
    try:
        tickets = [G.msm_device_buckets_async(h, d.data_ptr(), n, r, Nn) for r in range(min(Nn, 4))]
        parts = [G.msm_wait(t) for t in tickets] + [G.msm_wait(G.msm_device_buckets_async(h, d.data_ptr(), n, r, Nn)) for r in range(4, Nn)]
        okb = np.array_equal(G.g1_sum(np.stack(parts)), full)
        cuts = [W * n * r // Nn for r in range(Nn + 1)]
        parts = [G.msm_wait(G.msm_device_rows_async(h, d.data_ptr(), n, a, b)) for a, b in zip(cuts[:-1], cuts[1:])]
        okr = np.array_equal(G.g1_sum(np.stack(parts)), full)
    except BbGpuError:  # row / bucket shares are defined on one table segment
        assert segmented
        want = O.msm_affine(aligned_copy(sc), tab, n)
        okb = okr = (int(full[7]) >> 63) == (int(want[7]) >> 63) and ((int(want[7]) >> 63) or np.array_equal(full[:8], want[:8]))
    # point-range shares (bench.py --shard points): ranges of the whole table, up to four in flight (the throughput choices of msm_issue_batch)
    pc = [n * r // Nn for r in range(Nn + 1)]
    tickets, parts = [], []
    for a, b in zip(pc[:-1], pc[1:]):
        if b > a:
            tickets.append(G.msm_device_async(h, d.data_ptr() + a * 32, b - a, a))
        if len(tickets) == 4:
            parts.append(G.msm_wait(tickets.pop(0)))
    parts += [G.msm_wait(t) for t in tickets]
    okp = np.array_equal(G.g1_sum(np.stack(parts)), full)
    share_cases += 1
    if not (okb and okr and okp):
        share_bad += 1
        print("SHARE MISMATCH n=%d kind=%d N=%d buckets_ok=%s rows_ok=%s points_ok=%s" % (n, kind, Nn, okb, okr, okp), flush=True)
    G.srs_release(h)
print("share splits: %d cases (bucket, row and point-range shares, 1..8 ranks), %d mismatches" % (share_cases, share_bad), flush=True)
bad += share_bad

# concurrent callers (round 4): four host threads on the host-pointer entries (MSM + transform), results against precomputed oracle answers
import threading
th_sizes = [4096, 1000, 10000, 257]
th_sc = [aligned_copy(pool[rng.permutation(N)[:n]]) for n in th_sizes]
th_want = [O.msm_affine(sc, table, n) for sc, n in zip(th_sc, th_sizes)]
th_co = [O.random_scalars(900 + k, 1 << (9 + k)) for k in range(4)]
th_ntt = [O.ntt(c, "coset_fft") for c in th_co]
th_bad, th_calls = [], [0] * 4
t3 = time.time() + budget / 6


def hammer(i):
    while time.time() < t3:
        got = G.pippenger(th_sc[i], table, th_sizes[i])
        if not np.array_equal(got[:8], th_want[i][:8]):
            th_bad.append(("msm", i))
        if not np.array_equal(G.coset_fft(th_co[i].copy()), th_ntt[i]):
            th_bad.append(("ntt", i))
        th_calls[i] += 2


ths = [threading.Thread(target=hammer, args=(i,)) for i in range(4)]
for t in ths:
    t.start()
for t in ths:
    t.join()
print("concurrent callers: %d calls from 4 threads, %d mismatches" % (sum(th_calls), len(th_bad)), flush=True)
bad += len(th_bad)

# pipelined: 4 different scalar vectors, two and three in flight, against their one-at-a-time results, at 2^15 and 2^20
for lg in (15, 20):
    n = 1 << lg
    xs = np.random.default_rng(3).integers(0, 1 << 64, size=4, dtype=np.uint64); xs[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
    h = G.srs_generate(xs, n)
    ds = []
    for j in range(4):
        s = np.random.default_rng(100 + j).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
        ds.append(torch.from_numpy(s.view(np.int64)).cuda())
    ref = [G.msm_wait(G.msm_device_async(h, d.data_ptr(), n)) for d in ds]
    rounds = mism = 0
    t1 = time.time() + budget / 4
    while time.time() < t1:
        if time.time() > t_say:
            print("... pipelined 2^%d: %d rounds so far, %d differing" % (lg, rounds, mism), flush=True); t_say = time.time() + 60
        for depth in (2, 3):
            infl, out = [], []
            for k in range(12):
                infl.append((k % 4, G.msm_device_async(h, ds[k % 4].data_ptr(), n)))
                if len(infl) == depth:
                    j, t = infl.pop(0); out.append((j, G.msm_wait(t)))
            while infl:
                j, t = infl.pop(0); out.append((j, G.msm_wait(t)))
            rounds += 1
            for j, r in out:
                if not np.array_equal(r, ref[j]):
                    mism += 1
    print("pipelined 2^%d: %d rounds of 12 MSMs, %d results differing from the one-at-a-time result" % (lg, rounds, mism), flush=True)
    bad += mism
sys.exit(1 if bad else 0)
