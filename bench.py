#!/usr/bin/env python3
"""bench.py -- BN254 G1 MSM points/s (+ Fr NTT elements/s) at n = 2^20 on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one 2^20-point MSM (random 253-bit scalars against the resident synthetic SRS x^i G) through the C ABI of
libbbgpu.so with inputs already in HBM, result normalised on the host.  With N > 1 ranks the MSM is sharded over the
ranks (north star): every rank accumulates its share of the SAME MSM, partial sums (96 bytes per rank) are exchanged
with one RCCL all-gather and folded on every rank -> "strong" scaling.  The NTT leg (single GPU by design) is timed the
same way on rank 0's GPU and reported in the "ntt" object of the same JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF: the parent below
never imports torch or touches HIP, it starts N fresh child processes of this file with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set (one rank per GPU), lets rank 0's JSON line through on its own stdout, and exits non-zero
if any child did (the other ranks are then ended by their exact PIDs).  Under torch.distributed.run the ranks already
exist and the file runs as one of them.

Only the cpu_baseline leg touches oracle/ (the reference's own code compiled into oracle/_ref, timed on the host).
"""
import argparse
import json
import os
import sys
import time


def launch_ranks(argv):
    """Parent side of `bench.py --gpus N`: returns None when this process is itself a rank (N == 1, or a launcher already set
    WORLD_SIZE), else the exit code of the N-rank job it ran.  Nothing here may import torch or load the HIP library: a process
    that has initialised the GPU must not be the one that starts the ranks."""
    import signal
    import socket
    import subprocess
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=None)
    known, _ = ap.parse_known_args(argv)
    n = known.gpus or 1
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return None
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                BBGPU_BENCH_LAUNCHER_PID=str(os.getpid()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(n)]
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    print("bench.py: rank %d exited with %d, ending the other ranks" % (r, code), file=sys.stderr, flush=True)
                    for o in sorted(live):
                        procs[o].terminate()  # exact PIDs of the children started above
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


if __name__ == "__main__":
    _rc = launch_ranks(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

# before HIP is initialised: the MSM slot streams get hardware queues of their own (DESIGN_HISTORY.md 6).  A rehearsal that puts all ranks on ONE GPU (--same-device) shares
# that GPU's hardware queues between the ranks: 4 ranks x 16 queues oversubscribe them and the GPU time-slices the processes (13 ms per step instead of 1.5)
_ranks_on_one_gpu = int(os.environ.get("WORLD_SIZE", "1")) if "--same-device" in sys.argv else 1
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, 32 // _ranks_on_one_gpu)) if _ranks_on_one_gpu > 1 else "16")
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from barretenberg_amd import BbGpu  # noqa: E402
from barretenberg_amd.sharding import PartialSumExchange, StepClock, pipelined_steps  # noqa: E402

LOG2N = 20
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
SPLITMIX_GAMMA = 0x9E3779B97F4A7C15


def splitmix64(state0, count):
    """`count` outputs of splitmix64 started at `state0` (the synthetic-input generator), vectorised"""
    with np.errstate(over="ignore"):
        z = np.uint64(state0) + np.uint64(SPLITMIX_GAMMA) * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def raw_scalars(n, state0):
    """a scalar = 4 consecutive outputs as limbs 0..3 with limb 3 & 0x0fff...f (value < 2^252 < r), NOT yet in Montgomery form"""
    a = splitmix64(state0, 4 * n).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    return a


def limbs_of(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


SKEWED_KINDS = ("all_equal", "zero_one", "small")


def skewed_scalars(kind, n):
    """Scalar vectors (Montgomery form, as the boundary takes them) whose digit distributions are far from uniform -- what real witnesses
    look like: every scalar equal; scalars in {0, 1, -1}; values below 200.  Pure splitmix64 arithmetic, so the fixtures of
    tests/golden/msm_r3.json (the reference's results on exactly these vectors, tools/gen_golden_r3.py) need no data beside the seeds."""
    mont = lambda v: limbs_of(v * (1 << 256) % FR_MODULUS)
    if kind == "all_equal":
        k = raw_scalars(1, 0xA11E0A11E0A11E01)[0]
        return np.tile(mont(sum(int(x) << (64 * i) for i, x in enumerate(k))), (n, 1))
    if kind == "zero_one":
        lut = np.stack([mont(0), mont(1), mont(FR_MODULUS - 1)])
        return lut[(splitmix64(0x0123456789ABCDEF, n) % np.uint64(3)).astype(np.int64)]
    if kind == "small":
        lut = np.stack([mont(v) for v in range(200)])
        return lut[(splitmix64(0x5A11C0DE5A11C0DE, n) % np.uint64(200)).astype(np.int64)]
    raise ValueError(kind)


def to_montgomery_on_device(G, raw, dev):
    """x -> x * 2^256 mod r for a whole vector, on the GPU (one pointwise Montgomery product with 2^512 mod r); returns the device tensor"""
    n = raw.shape[0]
    d_raw = torch.from_numpy(raw.view(np.int64)).to(dev)
    rsq = torch.from_numpy(limbs_of(pow(2, 512, FR_MODULUS)).view(np.int64)).to(dev)
    d_rsq = rsq.repeat(n, 1).contiguous()
    d_out = torch.empty_like(d_raw)
    G.mul_device(d_out.data_ptr(), d_raw.data_ptr(), d_rsq.data_ptr(), n)
    torch.cuda.synchronize()
    return d_out


def lib_sha16():
    """first 16 hex digits of the SHA-256 of the libbbgpu.so this process loaded (ties a bench line to a binary)"""
    import hashlib
    from barretenberg_amd.bbgpu import library_path
    try:
        return hashlib.sha256(open(library_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def median_ms(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(min(ts))


def cpu_baseline(table, scalars, expect_xy, ntt_in, ntt_expect, table16, expect16_xy, ntt22_in, ntt22_expect):
    """Reference CPU path (oracle/_ref = the reference's own sources, x86-64 asm) timed on this box's host cores on the same inputs:
    SURVEY 8d's list -- 1-thread pippenger and all-cores batched_scalar_multiplications at 2^20, config 1 (2^16, bucket widths 12 and 15,
    bench_barretenberg.cpp:308-332), fft and coset_fft at 2^20 and 2^22 (:603-611).  Medians.  Falls back to our C port when the
    reference build / BMI2+ADX are unavailable.  The GPU results are compared with the reference's on the way."""
    from oracle.pyoracle import Oracle, Ref, aligned_copy
    n = scalars.shape[0]
    out = {}
    if Ref.available(True):
        threads = min(16, os.cpu_count() or 1)
        os.environ.setdefault("OMP_NUM_THREADS", str(threads))
        R = Ref(True)
        R.set_threads(threads)
        sc, tb = aligned_copy(scalars), aligned_copy(table)

        def timed(fn, reps):
            ts, r = [], None
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), r

        t1, _ = timed(lambda: R.pippenger(sc, tb, n), 1)  # the reference's serial pippenger(), 1 thread: ~2.4 s, once
        tb_s, rb = timed(lambda: R.batched_msm([sc], [tb])[0], 3)  # the entry the prover uses, all granted cores
        match = bool(np.array_equal(rb[:8], expect_xy))
        # config 1: 2^16 points, forced bucket widths 12 and 15
        m = 1 << 16
        sc16, tb16 = aligned_copy(scalars[:m]), aligned_copy(table16)
        c1 = {}
        for width in (12, 15):
            tw, r16 = timed(lambda: R.pippenger(sc16, tb16, m, width), 3)
            c1["width_%d_ms" % width] = tw * 1e3
            match = match and bool(np.array_equal(R_norm(R, r16)[:8], expect16_xy))
        t16b, _ = timed(lambda: R.batched_msm([sc16], [tb16])[0], 3)  # the plumbing config on all granted cores
        c1["batched_all_cores_ms"] = t16b * 1e3
        # transforms
        ntt = {}
        ntt_match = True
        for lg, src, expect in ((16, ntt_in[:1 << 16], None), (20, ntt_in, ntt_expect), (22, ntt22_in, ntt22_expect)):
            co = aligned_copy(src)
            R.prepare_domain(1 << lg)
            for kind in ("fft", "coset_fft"):
                def run():
                    co[...] = src
                    R.ntt_inplace(co, kind)
                tn, _ = timed(run, 3)
                ntt["2^%d %s" % (lg, kind)] = {"ms": tn * 1e3, "elements_per_s": (1 << lg) / tn}
                if expect is not None and kind in expect:
                    ntt_match = ntt_match and bool(np.array_equal(co, expect[kind]))
        out = {"value": n / tb_s, "unit": "points/s", "cores": threads, "kind": "reference",
               "sample": "medians: 2^20-point MSM 1x pippenger() on 1 thread (%.0f ms, %.3e points/s), 3x batched_scalar_multiplications() on %d threads (%.0f ms); "
                         "config 1 pippenger(2^16) width 12 / 15 on 1 thread %.0f / %.0f ms, batched on all cores %.1f ms; fft / coset_fft on %d threads 2^16 %.2f / %.2f ms, "
                         "2^20 %.1f / %.1f ms, 2^22 %.1f / %.1f ms (each includes the memcpy that restores the input)" % (
                             t1 * 1e3, n / t1, threads, tb_s * 1e3, c1["width_12_ms"], c1["width_15_ms"], c1["batched_all_cores_ms"], threads, ntt["2^16 fft"]["ms"],
                             ntt["2^16 coset_fft"]["ms"], ntt["2^20 fft"]["ms"], ntt["2^20 coset_fft"]["ms"], ntt["2^22 fft"]["ms"], ntt["2^22 coset_fft"]["ms"]),
               "single_thread_value": n / t1, "single_thread_ms": t1 * 1e3, "all_cores_ms": tb_s * 1e3, "config1_2e16": c1, "ntt": ntt,
               "ntt_value": ntt["2^20 fft"]["elements_per_s"], "gpu_result_bit_exact": match and ntt_match}
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


def R_norm(R, p):
    """normalised affine coordinates of a reference Jacobian result (checker only)"""
    return R.g1_op("normalize", p)


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
        # the drop-in path north_star names: the reference's UNMODIFIED prover, its hot-path symbols resolved by libbbshim.so -> libbbgpu.so
        # (oracle/_ref/plonk_gpu, a child process; third proof of that process = steady state), with the shim's own accounting of where the
        # time goes: caller_ms - inside_shim_ms is the reference's own host code
        exe_gpu = os.path.join(ROOT, "oracle", "_ref", "plonk_gpu")
        if os.path.exists(exe_gpu):
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".json") as tf:
                r2 = subprocess.run([exe_gpu, "prove", str(gates)], cwd=ROOT, capture_output=True, text=True,
                                    env=dict(os.environ, OMP_NUM_THREADS=str(threads), BB_WARM_PROOFS="2", BBGPU_SHIM_PROFILE=tf.name, BBGPU_SHIM_STRICT="1"))
                m2 = re.search(r"construct_proof ([0-9.]+) ms", r2.stderr)
                try:
                    prof = json.loads(open(tf.name).read().strip().split("\n")[0])
                except Exception:
                    prof = None
            if m2:
                out["reference_prover_on_shim"] = {"ms": float(m2.group(1)), "cores": threads, "proof_bit_exact": r2.stdout.strip().split("\n")[:26] == ref_lines[:26],
                                                   "inside_shim_ms": prof and prof.get("inside_shim_ms"), "reference_host_code_ms": prof and (prof["caller_ms"] - prof["inside_shim_ms"]),
                                                   "h2d_bytes": prof and prof.get("h2d_bytes"), "d2h_bytes": prof and prof.get("d2h_bytes"),
                                                   "note": "third proof of the child process (BB_WARM_PROOFS=2); breakdown per symbol of the builder's run: %s" % (os.path.relpath(newest_profile("shim_profile_2e16_third_proof.json") or newest_profile("shim_profile_2e16.json") or "profiles/", ROOT))}
    P.destroy()
    G.srs_release(srs)
    return out


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the newest round that has one (the committed evidence a live figure is printed beside), or "" """
    import glob
    import re
    best = (-1, "")
    for q in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d\d)_", os.path.basename(q))
        if m and int(m.group(1)) > best[0]:
            best = (int(m.group(1)), q)
    return best[1]


NTT_RAMP_MS = 120.0  # untimed transforms before the first timed one, see device_ntt_leg


def device_ntt_leg(G, dev, d_vec, n, kinds, steps, warmup, barrier):
    """in-place transforms on a device-resident vector; HIP events on the stream the kernels are launched on.
    Like the main leg (run_steps(40) before the W warm-up steps) the leg first leaves the post-idle ramp: after an idle period -- the upload of the
    vector, the host work before it -- the same transform takes 0.115 ms (2^20) / 0.50 ms (2^22) and falls to 0.099 / 0.43 ms over the next ~40 ms of
    back-to-back transforms (tools/ntt_ramp.py, profiles/r05_ntt_ramp.txt); W = 5 warm-up transforms are 0.6 ms.  NTT_RAMP_MS of untimed
    transforms come first, then the W warm-up transforms, then exactly `steps` timed ones."""
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    torch.cuda.synchronize()
    out = {}
    for kind in kinds:
        for _ in range(2):  # the first transform of a domain size / kind builds its tables: not part of the estimate below
            G.ntt_device(d_vec.data_ptr(), n, kind, stream=stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(tstream)
        for _ in range(8):
            G.ntt_device(d_vec.data_ptr(), n, kind, stream=stream)
        e1.record(tstream)
        torch.cuda.synchronize()
        ramp = int(NTT_RAMP_MS / max(1e-3, e0.elapsed_time(e1) / 8))
        for _ in range(ramp + warmup):
            G.ntt_device(d_vec.data_ptr(), n, kind, stream=stream)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(tstream)
        for _ in range(steps):
            G.ntt_device(d_vec.data_ptr(), n, kind, stream=stream)
        e1.record(tstream)
        barrier()
        wall = (time.perf_counter() - t0) / steps
        out[kind] = {"ms_per_step": wall * 1e3, "elements_per_s": n / wall, "device_ms": e0.elapsed_time(e1) / steps, "untimed_ramp_transforms": ramp + 10, "warmup": warmup}
    return out


def ntt_roofline(n, device_ms, traffic):
    b = 2 * 32 * n  # the vector read once, written once
    return {"bound": "hbm", "achieved": b / (device_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic, "kernel": "ntt_pass_fused_kernel x2", "algorithmic_bytes": b}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; N > 1 without a launcher starts the N ranks itself (launch_ranks)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2n", type=int, default=LOG2N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-window-tables", action="store_true", help="disable the pre-shifted SRS window tables")
    ap.add_argument("--no-plonk", action="store_true", help="skip the BASELINE config 5 leg (resident PLONK prover, 2^16 gates)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary-inclusive (host-pointer, PCIe) legs, the 2^22 transforms and config 1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the partial-sum exchange (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (1-GPU box, gloo)")
    ap.add_argument("--probe", action="store_true", help="no MSM: form the process group of the N ranks over --backend (nccl = RCCL on the ranks' GPUs), one all-gather of 96 bytes, "
                    "report who is there and on which devices, exit; non-zero with the backend's own error text when the group cannot be formed")
    ap.add_argument("--exchange-group", type=int, default=0, help="N > 1: partial sums of this many MSMs per all-gather (0 = the pipeline depth)")
    ap.add_argument("--issuer-thread", type=int, default=1, help="N > 1: 1 = the shares are issued by a helper thread while the main thread collects and exchanges")
    ap.add_argument("--no-window-split-leg", action="store_true", help="N > 1 with the default split: skip the extra timed leg that runs the same MSM split by window rows")
    ap.add_argument("--shard", default="points", choices=("points", "buckets", "rows", "windows"),
                    help="N > 1: how one MSM is split over the ranks (points: rank r holds points and scalars [n r / N, n (r + 1) / N) with all their digit windows -- "
                         "the reference's own per-thread slicing, and the one split where the digit kernel's work divides by N too; "
                         "rows: 1/N of the (window, point) table rows, all scalars on every rank; buckets: every rank reduces 1/N of the bucket range over all windows; "
                         "windows: whole digit windows)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is None:
        args.gpus = world
    if world != args.gpus:  # launch_ranks() makes them equal; a foreign launcher must agree with the flag
        print("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under torch.distributed.run with --nproc-per-node N" % (args.gpus, world),
              file=sys.stderr)
        sys.exit(2)
    if args.same_device:
        local_rank = 0
    def device_identity(i):
        """PCI address of cuda:i as (domain, bus, device), -1s when the build does not expose it"""
        try:
            p = torch.cuda.get_device_properties(i)
            return [int(getattr(p, "pci_domain_id", -1)), int(getattr(p, "pci_bus_id", -1)), int(getattr(p, "pci_device_id", -1))]
        except Exception:
            return [-1, -1, -1]

    def rccl_version():
        try:
            v = torch.cuda.nccl.version()
            return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
        except Exception:
            return None

    if world > 1 or args.probe:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            if args.backend == "nccl":
                have = torch.cuda.device_count()  # counting devices does not initialise the GPU
                if have <= local_rank:
                    raise RuntimeError("rank %d needs cuda:%d, this node shows %d GPU(s)" % (rank, local_rank, have))
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
        except Exception as exc:  # the backend's own words, and a non-zero exit: never a silent switch to another backend
            print("bench.py: rank %d could not join the %s process group of %d: %s: %s" % (rank, args.backend, world, type(exc).__name__, exc), file=sys.stderr, flush=True)
            sys.exit(3)
    if args.probe:
        # A two-second sanity line of the group the timed run would use, on the devices it would use: who is there (rank, pid, parent pid, PCI address
        # of the rank's GPU) through ONE all-gather of 96 bytes per rank -- the size and shape of the partial-sum exchange -- over args.backend.
        on_gpu = args.backend == "nccl"
        pdev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
        mine = torch.tensor([rank, os.getpid(), os.getppid()] + (device_identity(local_rank) if on_gpu else [-1, -1, -1]) + [0] * 6, dtype=torch.int64, device=pdev)
        seen = torch.zeros(world * 12, dtype=torch.int64, device=pdev)
        def gather():
            if on_gpu or hasattr(dist, "all_gather_into_tensor"):
                try:
                    dist.all_gather_into_tensor(seen, mine)
                    return
                except (RuntimeError, NotImplementedError):
                    if on_gpu:
                        raise
            dist.all_gather(list(seen.view(world, 12).unbind(0)), mine)  # a CPU backend without the flat form

        t0 = time.perf_counter()
        try:
            gather()
            if on_gpu:
                torch.cuda.synchronize()
        except Exception as exc:
            print("bench.py: rank %d: the 96-byte all-gather over %s failed: %s: %s" % (rank, args.backend, type(exc).__name__, exc), file=sys.stderr, flush=True)
            sys.exit(4)
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        for _ in range(20):
            gather()
        if on_gpu:
            torch.cuda.synchronize()
        each_us = (time.perf_counter() - t0) / 20 * 1e6
        seen = seen.cpu().view(world, 12)
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "backend": args.backend, "rccl_ranks": dist.get_world_size(), "rccl_version": rccl_version() if on_gpu else None,
                              "ranks": [int(t[0]) for t in seen], "pids": [int(t[1]) for t in seen], "parent_pids": [int(t[2]) for t in seen],
                              "devices": ["%04x:%02x:%02x" % (int(t[3]), int(t[4]), int(t[5])) if int(t[4]) >= 0 else None for t in seen],
                              "all_gather_96B_first_ms": first_ms, "all_gather_96B_us": each_us,
                              "launcher_pid": int(os.environ.get("BBGPU_BENCH_LAUNCHER_PID", "0"))}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        fail = os.environ.get("BBGPU_BENCH_PROBE_FAIL_RANK")
        sys.exit(7 if fail is not None and int(fail) == rank else 0)
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

    # ---- synthetic inputs, identical on every rank, resident in HBM before any timed region --------------
    # scalars: splitmix64 from state 0x9e3779b97f4a7c15, 4 outputs per scalar, limb 3 masked to 60 bits, converted to Montgomery form;
    # SRS: P_i = x^i G with x = the first scalar of a separately seeded stream
    x_raw = raw_scalars(1, 0x5EED0F5EC2E7C0DE)[0]
    x_secret = limbs_of(sum(int(v) << (64 * i) for i, v in enumerate(x_raw)) * (1 << 256) % FR_MODULUS)
    single = (rank == 0 and world == 1)
    cpu_leg = single and not args.no_cpu_baseline  # the CPU reference is timed at N = 1 only
    host_legs = single and not args.no_boundary   # boundary-inclusive legs need the host copy of the point table
    by_points = world > 1 and args.shard == "points" and not args.no_window_tables
    pt0, pt1 = (n * rank // world, n * (rank + 1) // world) if by_points else (0, n)  # this rank's point range
    if world > 1 and not args.no_window_tables and args.shard in ("rows", "windows"):
        G.set_table_share(rank, world)  # this rank's 1/N of the (window, point) rows touches ceil(W / N) + 1 digit windows: only those are built and kept
    if by_points:
        G.set_point_share(world)        # the slice's tables take the window size of the whole MSM
    torch.cuda.synchronize()
    t_srs0 = time.perf_counter()
    if cpu_leg or host_legs:
        srs, table = G.srs_generate(x_secret, n, want_host_table=True)
    else:
        srs, table = G.srs_generate(x_secret, pt1 - pt0, first=pt0), None  # a point-range rank generates and keeps its slice only
    torch.cuda.synchronize()
    srs_setup_ms = (time.perf_counter() - t_srs0) * 1e3  # one-time: points generated, window tables built (and the host copy written when asked for)
    G.set_table_share(0, 1)
    G.set_point_share(1)
    d_scalars = to_montgomery_on_device(G, raw_scalars(n, SPLITMIX_GAMMA), dev)
    scalars = d_scalars.cpu().numpy().view(np.uint64) if (cpu_leg or host_legs) else None
    W = G.srs_num_windows(srs, pt1 - pt0)
    wb, we = W * rank // world, W * (rank + 1) // world
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the 96-byte partial sums are exchanged

    # with window tables (one shared bucket set) the W * n (window, point) pairs split at ANY row, so every rank takes exactly 1/N of them
    # even when N does not divide W (15 windows of 17 bits at 2^20); without tables the split follows whole windows
    wb_tab, we_tab = (W * rank // world, -(-W * (rank + 1) // world)) if (world > 1 and args.shard in ("rows", "windows")) else (0, W)  # digit windows whose tables this rank keeps
    by_buckets = world > 1 and args.shard == "buckets" and G.srs_has_window_tables(srs)  # every rank over all windows and points, 1 / N of the bucket range
    by_rows = world > 1 and args.shard == "rows" and G.srs_has_window_tables(srs)
    rows = (W * n * rank // world, W * n * (rank + 1) // world)

    def issue():
        if by_points:
            return G.msm_device_async(srs, d_scalars.data_ptr() + pt0 * 32, pt1 - pt0) if pt1 > pt0 else None
        if by_buckets:
            return G.msm_device_buckets_async(srs, d_scalars.data_ptr(), n, rank, world)
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

    # shares in flight: two at N <= 2 (more changes nothing there: the step is the accumulation), four from N = 4 on, where a share is a chain of
    # short launches (tools/share_ab.py with GPU_MAX_HW_QUEUES=8, a middle 1/8 row share: 0.362 / 0.254 / 0.217 / 0.213 / 0.232 / 0.219 ms per step
    # with 1 / 2 / 3 / 4 / 6 / 8 in flight; 1/4: 0.589 / 0.372 / 0.350 / 0.348 / 0.351 / 0.359)
    depth = 4 if world >= 4 else 2
    # the partial sums of the `depth` shares in flight travel in ONE all-gather, and the shares are issued by a helper thread: a rank's host
    # thread paid 44 us issue + 15 us wait + 63 us exchange per step against a share of ~0.18 ms at N = 8 (profiles/r03_*_cost.txt)
    exchange = PartialSumExchange(G, world, xdev, group=(args.exchange_group or depth)) if world > 1 else None
    use_issuer = world > 1 and args.issuer_thread != 0
    clock = StepClock()

    def run_steps(k):
        """k complete MSMs; step i+1 is enqueued before step i is collected (two-slot pipeline of the library), so the
        bucket-reduction tail + host finish of one step overlap the sort/accumulate of the next; with N > 1 the exchange of
        step i is in flight while step i+1 is collected (barretenberg_amd/sharding.py)"""
        out = pipelined_steps(k, issue, collect, exchange, depth=depth, clock=clock, issuer=use_issuer, issuer_device=local_rank)
        return out[-1] if out else None

    def finish(ticket):
        part = collect(ticket)
        if world == 1:
            return part
        got = exchange.finish(exchange.start([part]))
        return got[0] if isinstance(got, list) else got

    # The first ~50 ms of sustained work after idle run ~4 % slower than the steady state on this part (tools/step_gap.py: 1.33 ms for the
    # first 20-step batch, 1.27 ms for every later one, independent of inputs and instrumentation), so the pipeline is brought to its
    # steady state before the W warm-up steps the caller asked for -- untimed, like them.
    run_steps(40)
    res = run_steps(args.warmup)
    # live timing of the dominant kernel INSIDE the timed region: the library brackets the accumulation with HIP events on the stream
    # it is launched on (timing level 2: two markers per MSM; an event after EVERY stage costs 0.085 ms per step in marker latency
    # on the launch chains and is taken in a separate, untimed pipelined run below)
    G.set_timing(2)
    stage_log.clear()
    barrier()
    clock.__init__()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    t_mine = time.perf_counter() - t0  # this rank's own wall time, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    G.set_timing(False)
    host_us = clock.per_step_us()
    acc_pipe = np.mean(np.array(stage_log), axis=0) if stage_log else np.zeros(8)
    G.set_timing(1)
    stage_log.clear()
    run_steps(max(6, args.steps // 2))
    barrier()
    G.set_timing(False)
    stage_pipe = np.mean(np.array(stage_log), axis=0) if stage_log else np.zeros(8)
    per_rank = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # one gather AFTER the timed region: every rank's own step time, the host time its thread(s) spent per step in issue / wait / exchange, its
        # GPU -- so that a disappointing curve can be read from the record instead of guessed at
        mine = torch.tensor([t_mine / args.steps * 1e3] + [host_us[k] for k in StepClock.KEYS] + [float(v) for v in device_identity(local_rank)],
                            dtype=torch.float64, device=xdev)
        allr = torch.zeros(world * mine.numel(), dtype=torch.float64, device=xdev)
        dist.all_gather_into_tensor(allr, mine) if exchange.flat else dist.all_gather(list(allr.view(world, -1).unbind(0)), mine)
        allr = allr.cpu().view(world, -1).numpy()
        per_rank = {"rank_ms_per_step": [float(v) for v in allr[:, 0]],
                    "rank_host_us_per_step": {k: [float(v) for v in allr[:, 1 + i]] for i, k in enumerate(StepClock.KEYS)},
                    "rank_skew_ms": float(allr[:, 0].max() - allr[:, 0].min()) * args.steps,
                    "devices": ["%04x:%02x:%02x" % (int(r[5]), int(r[6]), int(r[7])) if r[6] >= 0 else None for r in allr],
                    "rccl_version": rccl_version() if args.backend == "nccl" else None,
                    "exchange_group": exchange.group, "issuer_thread": bool(use_issuer), "shares_in_flight": depth,
                    "note": "rank_ms_per_step: each rank's own wall time over the timed steps / steps, before the closing barrier (ms_per_step is the MAX over ranks incl. the barrier); "
                            "rank_host_us_per_step: host time per step inside issue() (helper thread when issuer_thread), wait = collect of a share, exchange_start / _finish "
                            "(one all-gather per exchange_group steps); rank_skew_ms: (slowest - fastest rank) over the whole timed region"}
    msm_ms = dt / args.steps * 1e3
    # latency of one isolated MSM (no pipelining): median of 10 after 3
    msm_latency_ms, msm_latency_min = median_ms(lambda: finish(issue()), 10, 3)
    # what this rank holds for ITS share, before the full-size comparison runs below allocate theirs (N > 1: the workspace must be sized by the share)
    resident_after_timed = G.memory_stats()
    barrier()

    # window-sharded result == the same MSM done by one rank alone (outside the timed region)
    sharded_ok = None
    if world > 1:
        srs_full = G.srs_generate(x_secret, n)  # complete tables, for this check only
        full = G.msm_device(srs_full, d_scalars.data_ptr(), n)
        G.srs_release(srs_full)
        sharded_ok = bool(np.array_equal(full, res))

    # ---- N > 1, default (point-range) split: the north star's WINDOW split of the same MSM next to it -- every rank 1/N of the window-major (window, point)
    # rows of its own window tables (whole windows when N divides W), same pipeline, same exchange, timed the same way; reported as `window_split`
    window_split = None
    if by_points and not args.no_window_split_leg:
        G.set_table_share(rank, world)
        srs_rows = G.srs_generate(x_secret, n)
        G.set_table_share(0, 1)
        Wr = G.srs_num_windows(srs_rows, n)
        rr = (Wr * n * rank // world, Wr * n * (rank + 1) // world)

        def issue_rows():
            return G.msm_device_rows_async(srs_rows, d_scalars.data_ptr(), n, rr[0], rr[1]) if rr[1] > rr[0] else None

        def run_rows(k):
            out = pipelined_steps(k, issue_rows, collect, exchange, depth=depth, issuer=use_issuer, issuer_device=local_rank)
            return out[-1] if out else None

        run_rows(20 + args.warmup)
        barrier()
        t0 = time.perf_counter()
        res_rows = run_rows(args.steps)
        barrier()
        dt_rows = time.perf_counter() - t0
        tt = torch.tensor([dt_rows], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_rows = float(tt.item())
        window_split = {"what": "the same MSM split by table row = window-major (window, point) pairs, W n / N rows per rank (bench.py --shard rows), same steps / warm-up / exchange",
                        "ms_per_step": dt_rows / args.steps * 1e3, "value": n / (dt_rows / args.steps), "result_equals_point_split": bool(np.array_equal(res_rows, res))}
        G.srs_release(srs_rows)
        stage_log.clear()

    # ---- the same stages with nothing else on the GPU (one MSM at a time), for comparison ---------------------------
    G.set_timing(True)
    stage = np.zeros(7)
    reps = 5
    for _ in range(reps):
        if by_points:
            if pt1 > pt0:
                G.msm_device(srs, d_scalars.data_ptr() + pt0 * 32, pt1 - pt0)
                stage += np.array(G.last_timing()[:7])
        elif by_buckets:
            G.msm_wait(G.msm_device_buckets_async(srs, d_scalars.data_ptr(), n, rank, world))
            stage += np.array(G.last_timing()[:7])
        elif by_rows:
            G.msm_wait(G.msm_device_rows_async(srs, d_scalars.data_ptr(), n, rows[0], rows[1]))
            stage += np.array(G.last_timing()[:7])
        elif we > wb:
            G.msm_device(srs, d_scalars.data_ptr(), n, 0, wb, we)
            stage += np.array(G.last_timing()[:7])
    stage /= reps
    G.set_timing(False)

    # ---- NTT leg (single GPU; every rank runs it so the barrier semantics stay simple, rank 0 reports) ---------------
    ntt_in = raw_scalars(n, 0xF00D)
    d_co = torch.from_numpy(ntt_in.view(np.int64)).to(dev)
    ntt = device_ntt_leg(G, dev, d_co, n, ("fft", "coset_fft"), args.steps, args.warmup, barrier)
    ntt_out = None
    if cpu_leg:
        ntt_out = {}
        for kind in ("fft", "coset_fft"):
            d_chk = torch.from_numpy(ntt_in.view(np.int64)).to(dev)
            G.ntt_device(d_chk.data_ptr(), n, kind)
            torch.cuda.synchronize()
            ntt_out[kind] = d_chk.cpu().numpy().view(np.uint64)

    # ---- single-GPU extras (rank 0, N = 1): 4 * 2^20 transforms, config 1, the boundary-inclusive numbers ---------------------------
    ntt22 = ntt22_in = ntt22_out = None
    config1 = boundary = skewed = None
    table16 = res16 = None
    if host_legs:
        n22 = 1 << 22
        ntt22_in = raw_scalars(n22, 0xBEEF)
        d22 = torch.from_numpy(ntt22_in.view(np.int64)).to(dev)
        ntt22 = device_ntt_leg(G, dev, d22, n22, ("fft", "coset_fft"), max(5, args.steps // 2), args.warmup, barrier)
        if cpu_leg:
            ntt22_out = {}
            for kind in ("fft", "coset_fft"):
                d22.copy_(torch.from_numpy(ntt22_in.view(np.int64)))
                G.ntt_device(d22.data_ptr(), n22, kind)
                torch.cuda.synchronize()
                ntt22_out[kind] = d22.cpu().numpy().view(np.uint64)
        del d22
        # config 1 (bench_barretenberg.cpp:308-332): 2^16 points of the same SRS and the first 2^16 scalars
        m = 1 << 16
        table16 = table[:2 * m].copy()  # its OWN memory: registering a view of the big table would hand back (and later release) the big table's handle
        h16 = G.srs_register(table16)
        assert h16 != srs
        for _ in range(20):  # out of the post-idle ramp first, like the main leg (tools/step_gap.py)
            G.msm_device(h16, d_scalars.data_ptr(), m)
        lat16, lat16_min = median_ms(lambda: G.msm_device(h16, d_scalars.data_ptr(), m), 10, 3)

        def in_flight(depth, k=12):
            infl = []
            for _ in range(k):
                infl.append(G.msm_device_async(h16, d_scalars.data_ptr(), m))
                if len(infl) == depth:
                    G.msm_wait(infl.pop(0))
            while infl:
                G.msm_wait(infl.pop(0))
        pipe16, _ = median_ms(lambda: in_flight(2, 10), 10, 2)
        pipe16_3, _ = median_ms(lambda: in_flight(3, 12), 10, 2)  # small MSMs are chains of short launches: a third one in flight still fits (hardware queues: DESIGN_HISTORY.md 6 iv)
        res16 = G.msm_device(h16, d_scalars.data_ptr(), m)
        config1 = {"workload": "2^16-point G1 MSM (BASELINE config 1 on the GPU), inputs resident", "latency_ms": lat16, "latency_ms_min": lat16_min,
                   "ms_per_msm_two_in_flight": pipe16 / 10, "ms_per_msm_three_in_flight": pipe16_3 / 12, "points_per_s": m / (min(pipe16 / 10, pipe16_3 / 12) * 1e-3)}
        G.srs_release(h16)
        # skewed scalars at the full size (what real witnesses look like; parity of exactly these vectors against the reference's points:
        # tests/test_gpu_parity.py::test_msm_skewed_scalars_full_size): latency and two-in-flight step next to the uniform figures
        skewed = {"note": "2^20 points, inputs resident; latency = one MSM at a time (median of 5 after 2), step = two in flight (10 steps); uniform scalars: latency_ms_single_msm / ms_per_step"}
        for kind in SKEWED_KINDS:
            d_sk = torch.from_numpy(np.ascontiguousarray(skewed_scalars(kind, n)).view(np.int64)).to(dev)
            lat_sk, _ = median_ms(lambda: G.msm_device(srs, d_sk.data_ptr(), n), 5, 2)

            def two_sk(k=10):
                infl = []
                for _ in range(k):
                    infl.append(G.msm_device_async(srs, d_sk.data_ptr(), n))
                    if len(infl) == 2:
                        G.msm_wait(infl.pop(0))
                while infl:
                    G.msm_wait(infl.pop(0))
            step_sk, _ = median_ms(two_sk, 3, 1)
            skewed[kind] = {"latency_ms": lat_sk, "ms_per_step": step_sk / 10}
            del d_sk
        # boundary-inclusive: wall-clock around the drop-in calls with pageable host buffers -- H2D of the scalars (SRS
        # resident, as in the prover), H2D + D2H of the coefficients; median of 10 after 3; different scalars every call
        hs = [scalars, np.roll(scalars, 1, axis=0).copy(), np.roll(scalars, 2, axis=0).copy()]
        turn = [0]

        def one_msm():
            turn[0] += 1
            return G.pippenger(hs[turn[0] % 3], table, n)
        b_msm, b_msm_min = median_ms(one_msm, 10, 3)
        assert np.array_equal(G.pippenger(hs[0], table, n), res), "host-pointer MSM differs from the resident one"
        b_bat, b_bat_min = median_ms(lambda: G.batched_scalar_multiplications([(table, h, n) for h in hs]), 10, 3)
        # bbgpu_ntt is in place: the call ALONE is inside the timer, on a ring of three pre-filled pageable buffers (the refill happens outside it;
        # round 3 timed `refill + call` and subtracted a separately timed refill, which moved the figure by 20 % between boxes)
        def ntt_boundary(src, kind, reps=10, warm=3):
            ring = [src.copy() for _ in range(3)]
            ts = []
            for i in range(warm + reps):
                buf = ring[i % 3]
                buf[...] = src
                t0 = time.perf_counter()
                G.ntt(buf, kind)
                if i >= warm:
                    ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts)), float(min(ts))
        bn = {}
        for kind in ("fft", "coset_fft"):
            t_med, t_min = ntt_boundary(ntt_in, kind)
            bn[kind] = {"ms": t_med, "ms_min": t_min, "elements_per_s": n / (t_med * 1e-3)}
        t22, t22_min = ntt_boundary(ntt22_in, "fft")
        boundary = {"note": "wall-clock around the C-ABI calls with pageable host buffers, PCIe included; median of 10 after 3 warm-ups; SRS resident (its one-time upload excluded)",
                    "msm_g1_2e20": {"call": "bbgpu_msm_g1 (pippenger)", "ms": b_msm, "ms_min": b_msm_min, "points_per_s": n / (b_msm * 1e-3), "h2d_bytes": 32 * n},
                    "msm_g1_batch_3x2e20": {"call": "bbgpu_msm_g1_batch (batched_scalar_multiplications), 3 jobs", "ms_per_msm": b_bat / 3, "ms_per_msm_min": b_bat_min / 3,
                                            "points_per_s": 3 * n / (b_bat * 1e-3)},
                    "ntt_2e20": {"call": "bbgpu_ntt (fft / coset_fft)", "fft": bn["fft"], "coset_fft": bn["coset_fft"], "h2d_plus_d2h_bytes": 64 * n},
                    "ntt_2e22_fft": {"ms": t22, "ms_min": t22_min, "elements_per_s": n22 / (t22 * 1e-3)}}

    # ---- BASELINE config 5 (rank 0 only, reported beside the headline): the resident PLONK prover on a 2^16-gate circuit ----------
    plonk = None
    if single and not args.no_plonk:  # single-GPU leg; the N > 1 runs measure the window-sharded MSM only
        plonk = plonk_leg(G, args)

    if rank == 0:
        alg_bytes = n * (32 + 64) + 96  # every scalar and base point once, one result
        # average over the timed region's launches.  With two MSMs in flight the accumulation of step i+1 is ENQUEUED while that of step i
        # still runs, so the event pair around it also spans its wait in the queue (stage_pipe[3], kept as kernel_ms_events_raw); it cannot
        # execute before the previous one has drained, so its execution time is the spacing of consecutive end-of-accumulation events
        # when that is shorter (stage_pipe[7], HIP events on the launch streams as well: csrc/msm.hip finish_timing)
        acc_raw_ms = float(acc_pipe[3]) if acc_pipe[3] > 0 else float(stage[3])
        acc_ms = float(acc_pipe[7]) if acc_pipe[7] > 0 else acc_raw_ms
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes of the SAME binary (profiles/, tools/pmc_summary.py), corrected by the factor the
        # gather calibration kernel of known traffic gave for this access shape
        traffic = ntt_traffic = ntt22_traffic = None
        traffic_source = None
        tpath = newest_profile("pmc_traffic.json")
        if tpath:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("msm_accumulate_kernel_bytes_per_launch")
                ntt_traffic = tj.get("ntt_2e20_bytes_per_transform")
                ntt22_traffic = tj.get("ntt_2e22_bytes_per_transform")
                # NOT measured in this run: the counter passes need rocprofv3 around the process (tools/collect_profiles.sh); the figure is the committed one
                traffic_source = "%s (rocprofv3 --pmc passes on the builder's box, tools/collect_profiles.sh; library of that run: %s; this run's library: %s)" % (
                    os.path.relpath(tpath, ROOT), tj.get("library_sha256_16", "not recorded"), lib_sha16())
            except Exception:
                traffic = None
        # the same fraction from the committed rocprofv3 trace of this command (profiles/: spacing of consecutive accumulation dispatches'
        # end times, tools/acc_spacing.py), so that the live figure and the profile can be compared without reading profiles/README.md
        rocprof_spacing_ms = rocprof_frac = rocprof_spacing_file = None
        for sp in (newest_profile("acc_spacing.txt"),):
            if sp:
                import re
                rocprof_spacing_file = os.path.relpath(sp, ROOT)
                m = re.search(r"spacing[^0-9]*([0-9.]+) ms", open(sp).read())
                if m:
                    rocprof_spacing_ms = float(m.group(1))
                    rocprof_frac = alg_bytes / (rocprof_spacing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                break
        share_adds = (W * (pt1 - pt0)) if by_points else (W * n // world) if by_buckets else (rows[1] - rows[0]) if by_rows else n * (we - wb)  # mixed additions of this rank's accumulation (expected, for bucket shares)
        # the bound that does apply to the accumulation: instruction issue.  One mixed XYZZ addition (round-4 loop, ISA histogram tools/isa_hist.py: 2,048 VALU instructions
        # in the hot path + ~85 in the bucket-start block that ~1 trip in 4 runs) = 738 v_mad_u64_u32 with two VGPR factors + 753 with an SGPR factor (729 of the reductions,
        # 24 that add P / R / X3's addends inside them) + 144 v_lshrrev_b64 + 81 v_mul_lo_u32 + 170 v_and_b32 + ~162 other VALU, priced at the measured chip-wide issue rates of
        # tools/ubench/ubench_inst (445 / 489 / 565 / 537 / 916 / ~850 G wave-instructions/s) = the floor this instruction stream allows (round 3: 2,110 instructions, 4.04 ns)
        ns_per_wave_add = 738 / 445.0 + 753 / 489.0 + 144 / 565.0 + 81 / 537.0 + 170 / 916.0 + 162 / 850.0
        issue_floor_ms = share_adds / 64 * ns_per_wave_add * 1e-6
        line = {
            "metric": "BN254 G1 MSM points/sec at n=2^%d (Fr NTT elems/sec in 'ntt')" % args.log2n,
            "value": n / (msm_ms * 1e-3),
            "unit": "points/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "exchange_backend": (args.backend if world > 1 else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": msm_ms,
            "latency_ms_single_msm": msm_latency_ms,
            "latency_ms_single_msm_min": msm_latency_min,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32x9 (256-bit Montgomery, 29-bit limbs)",
            "data": "synthetic",
            "library_sha256_16": lib_sha16(),
            "value_boundary": (n / (boundary["msm_g1_2e20"]["ms"] * 1e-3)) if boundary is not None else None,  # points/s through the host-pointer drop-in call bbgpu_msm_g1, PCIe included
            "config": {"workload": "2^%d-point BN254 G1 MSM, splitmix64 scalars (< 2^252, Montgomery form) vs synthetic SRS x^i*G, inputs resident in HBM, result normalised" % args.log2n,
                       "one_time_costs": {"note": "paid once per SRS / per transform size, excluded from every timed figure",
                                          "srs_setup_ms": srs_setup_ms, "srs_setup_what": "bbgpu_srs_generate: points + window tables" + (" + host copy of the point table" if table is not None else ""),
                                          "srs_table_bytes": int(min(we_tab - wb_tab, W) * (pt1 - pt0) * 64) if not args.no_window_tables and W * n <= (1 << 24) else 0, "srs_points_bytes": (pt1 - pt0) * 64,
                                          "ntt_table_bytes": int(4 * n * 32) if n <= (1 << 22) else None,
                                          "resident_after_timed_region": resident_after_timed,  # the headline leg alone: SRS share + its workspaces (rank 0)
                                         "resident_now": G.memory_stats(),  # bbgpu_memory_stats at the end of the run: SRS points / window tables, NTT tables (all domain sizes used, under their byte budget), workspaces, pinned host memory
                                          },
                       "parallelism": ("%d digit windows x n points sharded %s over %d ranks, one all-gather of 96 B partial sums" % (W, "by point range (n / N points and scalars each, all windows)" if by_points else "by bucket range (all windows and points, 1 / N of the buckets each)" if by_buckets else "by table row (W n / N rows each)" if by_rows else "by window", world)) if world > 1 else "single GPU, %d digit windows of %d bits" % (W, -(-254 // W)),
                       "srs": "resident, with pre-shifted window tables" if not args.no_window_tables else "resident base points only"},
            "stage_ms": {"device_total": float(stage[0]), "digits": float(stage[1]), "sort": float(stage[2]), "accumulate": float(stage[3]),
                         "merge": float(stage[4]), "bucket_folds": float(stage[5]), "slices_collect": float(stage[6]),
                         "note": "one MSM at a time (no overlap)"},
            "stage_ms_in_timed_region": {"device_total": float(stage_pipe[0]), "digits": float(stage_pipe[1]), "sort": float(stage_pipe[2]),
                                         "accumulate": float(stage_pipe[3]), "merge": float(stage_pipe[4]), "bucket_folds": float(stage_pipe[5]),
                                         "slices_collect": float(stage_pipe[6]),
                                         "note": "two MSMs in flight, a separate run after the timed region with an event after every stage: stages of consecutive steps overlap, so they sum to more than ms_per_step"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": "msm_accumulate_kernel", "kernel_ms": acc_ms, "kernel_ms_events_raw": acc_raw_ms, "kernel_ms_alone": float(stage[3]),
                         "frac_from_rocprof_spacing": rocprof_frac, "rocprof_spacing_ms": rocprof_spacing_ms, "rocprof_spacing_source": rocprof_spacing_file,
                         "note": "integer-VALU bound (v_mad_u64_u32), not HBM bound: see DESIGN.md; algorithmic bytes %d per launch" % alg_bytes,
                         "valu": {"bound": "VALU instruction issue of the mixed addition's instruction stream at the measured per-instruction rates",
                                  "floor_ms": issue_floor_ms, "achieved_ms": acc_ms, "frac": issue_floor_ms / acc_ms if acc_ms > 0 else 0.0,
                                  "fq_multiplications_per_s": share_adds * 10 / (acc_ms * 1e-3) if acc_ms > 0 else 0.0,  # 8M + 2S per mixed addition (report fq mults/s)
                                  "mad_only": {"achieved": share_adds * 1467 / (acc_ms * 1e-3) / 1e12 if acc_ms > 0 else 0.0, "peak": 467e9 * 64 / 1e12,
                                               "unit": "T lane-mad/s", "frac": (share_adds * 1467 / (acc_ms * 1e-3)) / (467e9 * 64) if acc_ms > 0 else 0.0}}},
            "ntt": {"metric": "Fr radix-2 NTT elements/s at n=2^%d, in place on a device-resident vector" % args.log2n,
                    "fft": ntt["fft"], "coset_fft": ntt["coset_fft"], "roofline": ntt_roofline(n, ntt["fft"]["device_ms"], ntt_traffic)},
        }
        if ntt22 is not None:
            line["ntt_2e22"] = {"metric": "Fr radix-2 NTT elements/s at n=4*2^20 (BASELINE config 3)", "fft": ntt22["fft"], "coset_fft": ntt22["coset_fft"],
                                "roofline": ntt_roofline(1 << 22, ntt22["fft"]["device_ms"], ntt22_traffic)}
        if config1 is not None:
            line["config1_2e16"] = config1
            line["skewed_2e20"] = skewed
        if boundary is not None:
            line["boundary"] = boundary
        if sharded_ok is not None:
            line["sharded_result_equals_single_gpu"] = sharded_ok
        if window_split is not None:
            line["window_split"] = window_split
        if per_rank is not None:
            line.update(per_rank)
        if plonk is not None:
            line["plonk"] = plonk
        if cpu_leg:
            # expected outputs come from the reference run inside cpu_baseline; pass the GPU's so it can compare
            cb = cpu_baseline(table, scalars, res[:8], ntt_in, ntt_out, table16 if table16 is not None else table[:2 << 16],
                              res16[:8] if res16 is not None else G.msm_device(srs, d_scalars.data_ptr(), 1 << 16)[:8], ntt22_in if ntt22_in is not None else raw_scalars(1 << 22, 0xBEEF), ntt22_out)
            line["cpu_baseline"] = cb
            # what a drop-in caller sees (boundary-inclusive, one call at a time) against the reference's wall time for the same call
            if boundary is not None and "all_cores_ms" in cb:
                line["speedup_vs_cpu_all_cores"] = cb["all_cores_ms"] / boundary["msm_g1_2e20"]["ms"]
                line["speedup_vs_cpu_1_thread"] = cb["single_thread_ms"] / boundary["msm_g1_2e20"]["ms"]
            line["speedup_vs_cpu_all_cores_resident_pipelined"] = line["value"] / cb["value"]
            if "single_thread_value" in cb:
                line["speedup_vs_cpu_1_thread_resident_pipelined"] = line["value"] / cb["single_thread_value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    G.shutdown()


if __name__ == "__main__":
    main()
