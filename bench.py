#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM points/s (+ Fr NTT elements/s) at n = 2^20 on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one 2^20-point MSM (random 253-bit scalars against the resident synthetic SRS x^i G) through the C ABI of
libbbgpu.so with inputs already in HBM, result normalised on the host.  With N > 1 ranks the MSM's digit windows are
sharded over the ranks (north star): every rank accumulates its window range of the SAME MSM, partial sums (96 bytes
per rank) are exchanged with one RCCL all-gather and folded on every rank -> "strong" scaling.  The NTT leg (single
GPU by design) is timed the same way on rank 0's GPU and reported in the "ntt" object of the same JSON line.

Only the cpu_baseline leg touches oracle/ (the reference's own code compiled into oracle/_ref, timed on the host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from barretenberg_amd import BbGpu  # noqa: E402
from barretenberg_amd.sharding import PartialSumExchange, pipelined_steps  # noqa: E402

LOG2N = 20
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FR_TOP_MASK = 0x1FFFFFFFFFFFFFFF  # 253-bit values < r: uniformly random field elements read as Montgomery residues


def random_field_elements(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(FR_TOP_MASK)
    return a


def cpu_baseline(table, scalars, expect_xy, ntt_in, ntt_expect):
    """Reference CPU path (oracle/_ref = the reference's own sources, x86-64 asm) timed on this box's host cores on
    the same inputs.  Falls back to our C port when the reference build / BMI2+ADX are unavailable."""
    from oracle.pyoracle import Oracle, Ref, aligned_copy
    n = scalars.shape[0]
    out = {}
    if Ref.available(True):
        threads = min(16, os.cpu_count() or 1)
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        R = Ref(True)
        R.set_threads(threads)
        sc, tb = aligned_copy(scalars), aligned_copy(table)
        t0 = time.perf_counter()
        r1 = R.pippenger(sc, tb, n)  # the reference's serial pippenger(), 1 thread
        t1 = time.perf_counter() - t0
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            rb = R.batched_msm([sc], [tb])[0]  # the entry the prover uses, all granted cores
        tb_s = (time.perf_counter() - t0) / reps
        match = bool(np.array_equal(rb[:8], expect_xy))
        co = aligned_copy(ntt_in)
        R.prepare_domain(n)
        t0 = time.perf_counter()
        for _ in range(reps):
            co[...] = ntt_in
            R.ntt_inplace(co, "fft")
        tn = (time.perf_counter() - t0) / reps
        ntt_match = bool(np.array_equal(co, ntt_expect))
        out = {"value": n / tb_s, "unit": "points/s", "cores": threads, "kind": "reference",
               "sample": "2^20-point MSM: 1x pippenger() on 1 thread (%.0f ms, %.3e points/s) + %dx batched_scalar_multiplications() on %d threads (%.0f ms each); "
                         "2^20 fft() on %d threads %.1f ms = %.3e elements/s (includes a 32 MiB memcpy)" % (
                             t1 * 1e3, n / t1, reps, threads, tb_s * 1e3, threads, tn * 1e3, n / tn),
               "single_thread_value": n / t1, "ntt_value": n / tn, "gpu_result_bit_exact": match and ntt_match}
        del r1
    else:
        O = Oracle()
        m = 1 << 14
        t0 = time.perf_counter()
        r = O.msm_affine(aligned_copy(scalars[:m]), aligned_copy(table[:2 * m]), m)
        t = time.perf_counter() - t0
        out = {"value": m / t, "unit": "points/s", "cores": 1, "kind": "port",
               "sample": "2^14-point prefix of the workload through oracle/bn254_oracle.c (reference build unavailable here)",
               "gpu_result_bit_exact": None}
        del r
    return out


def plonk_leg(G, args, gates=65536, reps=10):
    """BASELINE config 5: construct_proof() of the resident prover (bbgpu_plonk_*) on the reference benchmark's add/mul-chain circuit
    (bench_plonk.cpp:25-37) of 2^16 gates; when the test-only reference build travelled with the repo and the cpu_baseline leg is on,
    the reference's own prover is timed on the host cores beside it and the two proofs are compared byte for byte."""
    import re
    import subprocess
    from barretenberg_amd.plonk import FR_MODULUS, Prover, bench_circuit, proof_lines, to_montgomery_limbs
    a0 = 0x0777777788888888555555556666666633333333444444441111111122222222
    b0 = 0x0ABCDEFABCDEFABC1234123412341234DDDDEEEEFFFF00009999AAAABBBBCCCC
    secret = 0x0123456789ABCDEF0F1E2D3C4B5A6978FEDCBA98765432100123456789ABCDEF
    state = bench_circuit(gates, a0, b0).preprocess()
    srs = G.srs_generate(to_montgomery_limbs([secret % FR_MODULUS])[0], state["n"])
    P = Prover(G, state, srs)
    first = P.construct_proof()
    prep = P.timing()["first_use_preparation_ms"]
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        P.construct_proof()
        ts.append((time.perf_counter() - t0) * 1e3)
    out = {"metric": "construct_proof ms, StandardComposer add/mul chain of %d gates, all polynomials resident" % gates, "n": int(state["n"]),
           "ms": float(np.median(ts)), "ms_min": float(min(ts)), "proofs_per_s": 1e3 / float(np.median(ts)),
           "circuit_only_preparation_ms": prep, "reference": None}
    exe = os.path.join(ROOT, "oracle", "_ref", "plonk_cpu")
    if not args.no_cpu_baseline and os.path.exists(exe) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "transcript.dat")):
        threads = min(16, os.cpu_count() or 1)
        r = subprocess.run([exe, "prove", str(gates)], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        m = re.search(r"construct_proof ([0-9.]+) ms", r.stderr)
        ref_lines = r.stdout.strip().split("\n")
        out["reference"] = {"kind": "reference", "cores": threads, "ms": float(m.group(1)) if m else None,
                            "proof_bit_exact": ref_lines[:26] == proof_lines(state["n"], first)}
        if m:
            out["speedup_vs_reference_cpu"] = float(m.group(1)) / out["ms"]
    P.destroy()
    G.srs_release(srs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=LOG2N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-window-tables", action="store_true", help="disable the pre-shifted SRS window tables")
    ap.add_argument("--no-plonk", action="store_true", help="skip the BASELINE config 5 leg (resident PLONK prover, 2^16 gates)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the partial-sum exchange (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box, gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.same_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    G = BbGpu(device=local_rank)
    G.set_precompute(not args.no_window_tables)
    n = 1 << args.log2n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic inputs, identical on every rank, resident in HBM before any timed region -------------------------
    x_secret = random_field_elements(1, 0x5EED)[0]
    cpu_leg = (rank == 0 and world == 1 and not args.no_cpu_baseline)  # the CPU reference is timed at N = 1 only
    want_table = cpu_leg
    if want_table:
        srs, table = G.srs_generate(x_secret, n, want_host_table=True)
    else:
        srs, table = G.srs_generate(x_secret, n), None
    scalars = random_field_elements(n, 0xC0FFEE)
    d_scalars = torch.from_numpy(scalars.view(np.int64)).to(dev)
    W = G.srs_num_windows(srs, n)
    wb, we = W * rank // world, W * (rank + 1) // world
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the 96-byte partial sums are exchanged

    # with window tables (one shared bucket set) the W * n (window, point) pairs split at ANY row, so every rank takes exactly 1/N of them
    # even when N does not divide W (15 windows of 17 bits at 2^20); without tables the split follows whole windows
    by_rows = world > 1 and G.srs_has_window_tables(srs)
    rows = (W * n * rank // world, W * n * (rank + 1) // world)

    def issue():
        if by_rows:
            return G.msm_device_rows_async(srs, d_scalars.data_ptr(), n, rows[0], rows[1])
        return G.msm_device_async(srs, d_scalars.data_ptr(), n, 0, wb, we) if we > wb else None

    stage_log = []  # per-MSM stage times (HIP events on the stream the kernels ran on), filled while timing is on

    def collect(ticket):
        """wait for this rank's MSM share; returns its normalised partial sum (96 bytes)"""
        if ticket is not None:
            part = G.msm_wait(ticket)
            tm = G.last_timing()
            if len(tm) >= 7:
                stage_log.append((tm + [tm[3]])[:8])  # [7]: accumulate without its time queued behind the previous accumulation
        else:
            part = np.zeros(12, dtype=np.uint64)
            part[7] = np.uint64(1 << 63)
        return part

    exchange = PartialSumExchange(G, world, xdev) if world > 1 else None

    def run_steps(k):
        """k complete MSMs; step i+1 is enqueued before step i is collected (two-slot pipeline of the library), so the
        bucket-reduction tail + host finish of one step overlap the sort/accumulate of the next; with N > 1 the exchange of
        step i is in flight while step i+1 is collected (barretenberg_amd/sharding.py)"""
        out = pipelined_steps(k, issue, collect, exchange)
        return out[-1] if out else None

    def finish(ticket):
        part = collect(ticket)
        return part if world == 1 else exchange.finish(exchange.start(part))

    res = run_steps(args.warmup)
    # live per-kernel timing INSIDE the timed region: the library brackets every stage with HIP events on the stream the
    # kernels are launched on (two event records per stage; the kernels themselves are unchanged)
    G.set_timing(True)
    stage_log.clear()
    barrier()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    G.set_timing(False)
    stage_pipe = np.mean(np.array(stage_log), axis=0) if stage_log else np.zeros(8)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    msm_ms = dt / args.steps * 1e3
    # latency of one isolated MSM (no pipelining), for the record
    t0 = time.perf_counter()
    for _ in range(5):
        finish(issue())
    barrier()
    msm_latency_ms = (time.perf_counter() - t0) / 5 * 1e3

    # window-sharded result == the same MSM done by one rank alone (outside the timed region)
    sharded_ok = None
    if world > 1:
        full = G.msm_device(srs, d_scalars.data_ptr(), n, 0, 0, W)
        sharded_ok = bool(np.array_equal(full, res))

    # ---- the same stages with nothing else on the GPU (one MSM at a time), for comparison ---------------------------
    G.set_timing(True)
    stage = np.zeros(7)
    reps = 5
    for _ in range(reps):
        if we > wb:
            G.msm_device(srs, d_scalars.data_ptr(), n, 0, wb, we)
            stage += np.array(G.last_timing()[:7])
    stage /= reps
    G.set_timing(False)

    # ---- NTT leg (single GPU; every rank runs it so the barrier semantics stay simple, rank 0 reports) ---------------
    ntt_in = random_field_elements(n, 0xF00D)
    d_co = torch.from_numpy(ntt_in.view(np.int64)).to(dev)
    tstream = torch.cuda.Stream(device=dev)  # the NTT kernels are launched on this stream, and so are the timing events
    stream = tstream.cuda_stream
    torch.cuda.synchronize()
    ntt = {}
    for kind in ("fft", "coset_fft"):
        for _ in range(args.warmup):
            G.ntt_device(d_co.data_ptr(), n, kind, stream=stream)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(tstream)
        for _ in range(args.steps):
            G.ntt_device(d_co.data_ptr(), n, kind, stream=stream)
        e1.record(tstream)
        barrier()
        wall = (time.perf_counter() - t0) / args.steps
        ntt[kind] = {"ms_per_step": wall * 1e3, "elements_per_s": n / wall, "device_ms": e0.elapsed_time(e1) / args.steps}
    d_chk = torch.from_numpy(ntt_in.view(np.int64)).to(dev)
    G.ntt_device(d_chk.data_ptr(), n, "fft", stream=stream)
    torch.cuda.synchronize()
    ntt_out = d_chk.cpu().numpy().view(np.uint64)

    # ---- BASELINE config 5 (rank 0 only, reported beside the headline): the resident PLONK prover on a 2^16-gate circuit ----------
    plonk = None
    if rank == 0 and world == 1 and not args.no_plonk:  # single-GPU leg; the N > 1 runs measure the window-sharded MSM only
        plonk = plonk_leg(G, args)

    if rank == 0:
        alg_bytes = n * (32 + 64) + 96  # every scalar and base point once, one result
        # average over the timed region's launches.  With two MSMs in flight the accumulation of step i+1 is ENQUEUED while that of step i
        # still runs, so the event pair around it also spans its wait in the queue (stage_pipe[3], kept as kernel_ms_events_raw); it cannot
        # execute before the previous one has drained, so its execution time is the spacing of consecutive end-of-accumulation events
        # when that is shorter (stage_pipe[7], HIP events on the launch streams as well: csrc/msm.hip finish_timing)
        acc_raw_ms = float(stage_pipe[3]) if stage_pipe[3] > 0 else float(stage[3])
        acc_ms = float(stage_pipe[7]) if stage_pipe[7] > 0 else acc_raw_ms
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("msm_accumulate_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        ntt_bytes = 2 * 32 * n
        share_adds = (rows[1] - rows[0]) if by_rows else n * (we - wb)  # mixed additions of this rank's accumulation
        line = {
            "metric": "BN254 G1 MSM points/sec at n=2^%d (Fr NTT elems/sec in 'ntt')" % args.log2n,
            "value": n / (msm_ms * 1e-3),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": msm_ms,
            "latency_ms_single_msm": msm_latency_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32x9 (256-bit Montgomery, 29-bit limbs)",
            "data": "synthetic",
            "config": {"workload": "2^%d-point BN254 G1 MSM, uniformly random 253-bit scalars vs synthetic SRS x^i*G, inputs resident in HBM, result normalised" % args.log2n,
                       "parallelism": ("%d digit windows x n points sharded %s over %d ranks, one all-gather of 96 B partial sums" % (W, "by table row (W n / N rows each)" if by_rows else "by window", world)) if world > 1 else "single GPU, %d digit windows of %d bits" % (W, -(-254 // W)),
                       "srs": "resident, with pre-shifted window tables" if not args.no_window_tables else "resident base points only"},
            "stage_ms": {"device_total": float(stage[0]), "digits": float(stage[1]), "sort": float(stage[2]), "accumulate": float(stage[3]),
                         "merge": float(stage[4]), "bucket_folds": float(stage[5]), "slices_collect": float(stage[6]),
                         "note": "one MSM at a time (no overlap)"},
            "stage_ms_in_timed_region": {"device_total": float(stage_pipe[0]), "digits": float(stage_pipe[1]), "sort": float(stage_pipe[2]),
                                         "accumulate": float(stage_pipe[3]), "merge": float(stage_pipe[4]), "bucket_folds": float(stage_pipe[5]),
                                         "slices_collect": float(stage_pipe[6]),
                                         "note": "two MSMs in flight: stages of consecutive steps overlap, so they sum to more than ms_per_step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "msm_accumulate_kernel", "kernel_ms": acc_ms, "kernel_ms_events_raw": acc_raw_ms, "kernel_ms_alone": float(stage[3]),
                         "note": "integer-VALU bound (v_mad_u64_u32), not HBM bound: see DESIGN.md; algorithmic bytes %d per launch" % alg_bytes,
                         # the bound that does apply: one mixed XYZZ addition per (point, window) = 1,467 v_mad_u64_u32 per lane,
                         # against the chip's measured issue rate for that instruction (470 G wave-instructions/s)
                         "valu": {"bound": "v_mad_u64_u32 issue", "achieved": share_adds * 1467 / (acc_ms * 1e-3) / 1e12 if acc_ms > 0 else 0.0,
                                  "peak": 470e9 * 64 / 1e12, "unit": "T lane-mad/s",
                                  "frac": (share_adds * 1467 / (acc_ms * 1e-3)) / (470e9 * 64) if acc_ms > 0 else 0.0}},
            "ntt": {"metric": "Fr radix-2 NTT elements/s at n=2^%d, in place on a device-resident vector" % args.log2n,
                    "fft": ntt["fft"], "coset_fft": ntt["coset_fft"],
                    "roofline": {"bound": "hbm", "achieved": ntt_bytes / (ntt["fft"]["device_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ntt_bytes / (ntt["fft"]["device_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "kernel": "ntt_pass_kernel x2"}},
        }
        if sharded_ok is not None:
            line["sharded_result_equals_single_gpu"] = sharded_ok
        if plonk is not None:
            line["plonk"] = plonk
        if cpu_leg:
            # expected NTT output comes from the reference run inside cpu_baseline; pass the GPU's so it can compare
            cb = cpu_baseline(table, scalars, res[:8], ntt_in, ntt_out)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_all_cores"] = line["value"] / cb["value"]
            if "single_thread_value" in cb:
                line["speedup_vs_cpu_1_thread"] = line["value"] / cb["single_thread_value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    G.shutdown()


if __name__ == "__main__":
    main()
