"""N > 1 path on CPU: world_size-2 gloo ranks exchange per-rank partial G1 sums (the MSM's one exchange step) and fold
them with the library's host-side bbgpu_g1_sum -- the same code bench.py runs over RCCL.  Partial sums come from the
oracle here (no GPU in this container): rank r owns half of the 16-bit digit windows of every scalar."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barretenberg_amd import BbGpu
    from oracle.pyoracle import FR, FR_MODULUS, Oracle, from_int, to_int
    O = Oracle()
    lib = BbGpu(init=False)
    x = O.random_scalars(0x5EED0F5EC2E7C0DE, 1)[0]
    srs = O.make_srs(x, n)
    scalars = O.random_scalars(0x9E3779B97F4A7C15, n)
    W, c = 16, 16
    wb, we = W * rank // world, W * (rank + 1) // world
    # partial sum over this rank's windows: sum_i (k_i restricted to bits [16 wb, 16 we)) * P_i
    lo, hi = c * wb, c * we
    acc = np.zeros(12, dtype=np.uint64)
    acc[7] = np.uint64(1 << 63)
    for i in range(n):
        k = to_int(O.from_mont(FR, scalars[i]))
        part = ((k >> lo) & ((1 << (hi - lo)) - 1)) << lo
        if part:
            acc = O.g1_add(acc, O.g1_scalar_mul(srs[i], O.to_mont(FR, from_int(part % FR_MODULUS))))
    mine = torch.from_numpy(O.g1_normalize_or_inf(acc).view(np.int64).copy())
    bufs = [torch.empty(12, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bufs, mine)
    total = lib.g1_sum(torch.stack(bufs).numpy().view(np.uint64))
    table = O.point_table(srs)
    want = O.msm_affine(scalars, table, n)
    ok = bool(np.array_equal(total[:8], want[:8]))
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(int(t.item()))
    dist.destroy_process_group()


def test_window_sharded_partial_sums_fold_over_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 24, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1


def _point_range_worker(rank, world, port, n, ret):
    """the split bench.py uses by default for N > 1: rank r holds points and scalars [n r / N, n (r + 1) / N) with all their digit windows (the reference's
    own slicing, scalar_multiplication.cpp:703-738); its partial sum here is the oracle's MSM over that range, the fold is the library's bbgpu_g1_sum"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barretenberg_amd import BbGpu
    from oracle.pyoracle import Oracle, aligned_copy
    O = Oracle()
    lib = BbGpu(init=False)
    x = O.random_scalars(0x5EED0F5EC2E7C0DE, 1)[0]
    table = O.point_table(O.make_srs(x, n))
    scalars = O.random_scalars(0x9E3779B97F4A7C15, n)
    a, b = n * rank // world, n * (rank + 1) // world  # uneven when world does not divide n
    part = np.zeros(12, dtype=np.uint64)
    part[7] = np.uint64(1 << 63)  # an empty range contributes the point at infinity
    if b > a:
        part = O.msm_affine(aligned_copy(scalars[a:b]), aligned_copy(table[2 * a:2 * b]), b - a)
    mine = torch.from_numpy(part.view(np.int64).copy())
    bufs = [torch.empty(12, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bufs, mine)
    total = lib.g1_sum(torch.stack(bufs).numpy().view(np.uint64))
    want = O.msm_affine(scalars, table, n)
    ok = bool(np.array_equal(total[:8], want[:8]))
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(int(t.item()))
    dist.destroy_process_group()


def test_point_range_partial_sums_fold_over_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_point_range_worker, args=(r, 2, port, 37, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1


def _pipeline_worker(rank, world, port, ret):
    """the step loop bench.py runs for N > 1 (barretenberg_amd/sharding.py) with the GPU work replaced by oracle points: step s of rank r
    contributes (s + 1) * (r + 2) * G; the folded result of step s must be (s + 1) * (2 + 3) * G, in step order, on every rank"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barretenberg_amd import BbGpu
    from barretenberg_amd.sharding import PartialSumExchange, pipelined_steps
    from oracle.pyoracle import FR, Oracle, from_int
    O = Oracle()
    lib = BbGpu(init=False)
    g = O.g1_one_affine()

    def mul(k):
        return O.g1_normalize_or_inf(O.g1_scalar_mul(g, O.to_mont(FR, from_int(k)))) if k else np.array([0] * 7 + [1 << 63] + [0] * 4, dtype=np.uint64)

    steps, issued = 7, []

    def issue():
        issued.append(len(issued))
        return issued[-1]

    def collect(ticket):
        return mul((ticket + 1) * (rank + 2) if not (rank == 1 and ticket == 3) else 0)  # one empty share (infinity) in the middle

    got = pipelined_steps(steps, issue, collect, PartialSumExchange(lib, world, torch.device("cpu")))
    ok = len(got) == steps
    for s in range(steps):
        want = mul((s + 1) * 2 + ((s + 1) * 3 if s != 3 else 0))
        ok = ok and bool(np.array_equal(got[s][:8], want[:8]))
    # the coalesced exchange bench.py uses for N > 1: the partial sums of `group` steps in ONE all-gather of group x 96 bytes (7 steps in groups
    # of 3, 3, 1: the tail group is padded with infinity), every MSM still folded from its own N partial sums, results in step order; with the
    # shares issued by the helper thread (issuer=True) and the per-step host clock bench.py reports
    from barretenberg_amd.sharding import StepClock
    for group, issuer, depth in ((3, False, 2), (3, True, 4), (4, True, 4), (2, True, 1)):
        issued.clear()
        clock = StepClock()
        got = pipelined_steps(steps, issue, collect, PartialSumExchange(lib, world, torch.device("cpu"), group=group), depth=depth, clock=clock, issuer=issuer)
        ok = ok and len(got) == steps and issued == list(range(steps)) and clock.steps == steps
        ok = ok and set(clock.per_step_us()) == {"issue", "wait", "exchange_start", "exchange_finish"} and clock.per_step_us()["exchange_start"] > 0
        for s in range(steps):
            want = mul((s + 1) * 2 + ((s + 1) * 3 if s != 3 else 0))
            ok = ok and bool(np.array_equal(got[s][:8], want[:8]))
    # an exception on the issuing thread reaches the caller
    def bad_issue():
        if len(issued) == 2:
            raise RuntimeError("issue failed")
        return issue()
    # ... and every ticket issued before the failure is still collected (they used to stay in the queue -- MSM slots left pending), with and
    # without the helper thread, at depths where one or two tickets are waiting when the failure arrives
    for issuer, depth in ((True, 2), (True, 4), (False, 2), (False, 3)):
        issued.clear()
        collected = []
        try:
            pipelined_steps(steps, bad_issue, lambda t: (collected.append(t), mul(t + 1))[1], None, depth=depth, issuer=issuer)
            ok = False
        except RuntimeError:
            pass
        ok = ok and sorted(collected) == issued == [0, 1]
    # world == 1 semantics: no exchange, partial sums come back as they are
    issued.clear()
    solo = pipelined_steps(3, issue, lambda t: mul(t + 1), None)
    ok = ok and len(solo) == 3
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret.put(int(t.item()))
    dist.destroy_process_group()


def test_pipelined_async_exchange_over_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert ret.get(timeout=10) == 1


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` (the driver's command form) with no launcher around it: the parent starts N fresh rank processes before
    anything touches torch / HIP, the ranks form ONE process group of size N, rank 0's line comes through the parent's stdout"""
    r, lines = _run_bench(["--gpus", "3", "--probe", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1
    p = lines[0]
    assert p["n_gpus"] == 3 and p["rccl_ranks"] == 3 and p["ranks"] == [0, 1, 2]
    assert len(set(p["pids"])) == 3 and p["launcher_pid"] not in p["pids"]
    assert p["parent_pids"] == [p["launcher_pid"]] * 3  # children of the launcher, not re-executions of it


def test_bench_launcher_reports_a_failed_rank():
    r, lines = _run_bench(["--gpus", "2", "--probe", "--backend", "gloo"], {"BBGPU_BENCH_PROBE_FAIL_RANK": "1"})
    assert r.returncode == 7 and "rank 1 exited with 7" in r.stderr


def test_bench_launcher_with_eight_ranks_and_one_that_fails():
    """the driver's largest form, rehearsed on the CPU (a one-GPU box may hold at most 6 processes on its card, so the 8-rank case never runs on one): 8
    rank processes form ONE gloo group of 8; when rank 5 exits non-zero the launcher ends the other seven and returns its code"""
    import time
    r, lines = _run_bench(["--gpus", "8", "--probe", "--backend", "gloo"])
    assert r.returncode == 0, r.stderr
    assert len(lines) == 1 and lines[0]["n_gpus"] == 8 and lines[0]["rccl_ranks"] == 8 and lines[0]["ranks"] == list(range(8))
    pids = lines[0]["pids"]
    assert len(set(pids)) == 8
    t0 = time.time()
    r, lines = _run_bench(["--gpus", "8", "--probe", "--backend", "gloo"], {"BBGPU_BENCH_PROBE_FAIL_RANK": "5"})
    assert r.returncode == 7 and "rank 5 exited with 7" in r.stderr  # (rank 0 may have printed its probe line before rank 5 died: the exit code is what tells)
    assert time.time() - t0 < 120  # the survivors were ended, not waited for until a collective times out
    import subprocess
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus 8 --probe"], capture_output=True, text=True).stdout.split()
    assert not left, left


def test_bench_refuses_a_world_size_that_contradicts_the_flag():
    r, lines = _run_bench(["--gpus", "4", "--probe", "--backend", "gloo"], {"WORLD_SIZE": "2", "RANK": "0", "MASTER_PORT": "29999"})
    assert r.returncode == 2 and not lines and "WORLD_SIZE=2" in r.stderr


def test_bench_under_torch_distributed_run():
    """the driver's N > 1 form: torch.distributed.run starts the ranks, bench.py must not start more"""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    port = 32500 + (os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--probe", "--backend", "gloo"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2 and lines[0]["launcher_pid"] == 0


def test_bench_probe_over_nccl_fails_loudly_without_gpus():
    """`--probe --backend nccl` (the default backend) is the RCCL sanity line of an N > 1 run: on a machine that cannot form the group it exits
    non-zero with the reason -- it never switches to gloo behind the caller's back"""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r, lines = _run_bench(["--gpus", "2", "--probe"])
    assert r.returncode != 0 and not lines
    assert "could not join the nccl process group" in r.stderr and "GPU" in r.stderr
