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
    # world == 1 semantics: no exchange, partial sums come back as they are
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
