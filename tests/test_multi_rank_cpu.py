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
