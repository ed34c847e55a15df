#!/usr/bin/env python3
"""Host time of one partial-sum exchange step (start + finish) over backend nccl, world_size 1 (the only RCCL configuration a one-GPU box
offers: the collective degenerates to a device copy, everything around it -- buffers, H2D, launch, D2H, host fold -- is what N ranks pay
too), the preallocated ring (barretenberg_amd/sharding.py) against the per-step allocation it replaced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from barretenberg_amd import BbGpu
from barretenberg_amd.sharding import PartialSumExchange

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
G = BbGpu(0)
dev = torch.device("cuda", 0)
part = np.zeros(12, dtype=np.uint64); part[7] = np.uint64(1 << 63)  # infinity: the fold is trivial, the plumbing is what is timed


def old_step():
    mine = torch.from_numpy(part.view(np.int64).copy()).to(dev)
    bufs = [torch.empty(12, dtype=torch.int64, device=dev) for _ in range(8)]  # an 8-rank step allocates 8
    work = dist.all_gather(bufs[:1], mine, async_op=True)
    work.wait()
    return G.g1_sum(torch.stack(bufs[:1]).cpu().numpy().view(np.uint64))


ex = PartialSumExchange(G, 1, dev)
new_step = lambda: ex.finish(ex.start(part))
ex4 = PartialSumExchange(G, 1, dev, group=4)
four = [part] * 4
for name, fn, per in (("per-step allocation (round 1)", old_step, 1), ("preallocated ring", new_step, 1),
                      ("one all-gather of 4 x 96 B per 4 steps (round 4)", lambda: ex4.finish(ex4.start(four)), 4)):
    for _ in range(50):
        fn()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        ts.append((time.perf_counter() - t0) / 200)
    print("%-50s %.1f us per exchange = %.1f us per MSM step (median of 5 x 200)" % (name, float(np.median(ts)) * 1e6, float(np.median(ts)) * 1e6 / per), flush=True)
dist.destroy_process_group()
