#!/usr/bin/env python3
"""What a rank's host side pays per step of an N = 8 run, end to end: the step loop bench.py runs for N > 1 (barretenberg_amd/sharding.py
pipelined_steps: issue a 1/8 point-range share, collect it, exchange the partial sums, fold) over backend nccl with world_size 1 -- the only RCCL
configuration a one-GPU box offers: the collective degenerates to a device copy, everything around it is what 8 ranks pay too -- with
  * one all-gather per step (round 3) against one per `group` = 4 steps (the shares in flight),
  * the shares issued by the collecting thread against a helper thread,
and the share alone (no exchange at all) as the floor.  Prints ms per step and the host microseconds per step inside issue / wait / exchange."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from barretenberg_amd import BbGpu
from barretenberg_amd.sharding import PartialSumExchange, StepClock, pipelined_steps

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
G = BbGpu(0)
dev = torch.device("cuda", 0)
n, N = 1 << 20, 8
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
m, off = n // N, (n // N) * (N // 2)
G.set_point_share(N)
srs = G.srs_generate(x, m, first=off)
G.set_point_share(1)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
issue = lambda: G.msm_device_async(srs, d.data_ptr() + off * 32, m)
collect = lambda t: G.msm_wait(t)
depth, steps = 4, 200
print("a middle 1/8 point-range share of a 2^20-point MSM, %d in flight, %d steps x 5 (median), backend nccl world_size 1" % (depth, steps))
for name, group, issuer in (("share alone, no exchange", 0, False), ("share alone, no exchange, issuer thread", 0, True),
                            ("exchange every step (round 3)", 1, False), ("exchange every step, issuer thread", 1, True),
                            ("one exchange per 4 steps", 4, False), ("one exchange per 4 steps, issuer thread", 4, True)):
    ex = PartialSumExchange(G, 1, dev, group=group) if group else None
    pipelined_steps(60, issue, collect, ex, depth=depth, issuer=issuer)
    torch.cuda.synchronize()
    ts, clock = [], StepClock()
    for _ in range(5):
        t0 = time.perf_counter()
        pipelined_steps(steps, issue, collect, ex, depth=depth, clock=clock, issuer=issuer)
        ts.append((time.perf_counter() - t0) / steps)
    us = clock.per_step_us()
    serial = us["issue"] + us["exchange_start"] + us["exchange_finish"]  # wait contains the GPU's time: not host work
    print("%-42s %.4f ms per step | host us per step: issue %.1f, wait (incl. the GPU) %.1f, exchange start %.1f + finish %.1f | launches + exchange %.1f us"
          % (name, float(np.median(ts)) * 1e3, us["issue"], us["wait"], us["exchange_start"], us["exchange_finish"], serial), flush=True)
dist.destroy_process_group()
