"""Window-sharded MSM across ranks: the one exchange step and the pipelined step loop bench.py runs.

Rank g computes the partial sum of its digit windows (bbgpu_msm_g1_device_async with a window range); the partial sums
(96 bytes per rank, normalised) are all-gathered -- RCCL over xGMI on the GPUs, gloo in the CPU tests; RCCL has no G1
reduction operator, so "all-reduce" = all-gather + the identical host fold bbgpu_g1_sum on every rank.  The exchange of
step i is asynchronous and overlaps the collection of step i + 1.
"""
import numpy as np
import torch
import torch.distributed as dist


class PartialSumExchange:
    def __init__(self, lib, world, device):
        """lib: BbGpu (only g1_sum is used: host arithmetic); device: where the 96-byte tensors live (cuda for nccl, cpu for gloo)"""
        self.lib, self.world, self.device = lib, world, device

    def start(self, part):
        mine = torch.from_numpy(np.ascontiguousarray(part, dtype=np.uint64).view(np.int64).copy()).to(self.device)
        bufs = [torch.empty(12, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        return dist.all_gather(bufs, mine, async_op=True), bufs, mine

    def finish(self, handle):
        work, bufs, _ = handle
        work.wait()
        return self.lib.g1_sum(torch.stack(bufs).cpu().numpy().view(np.uint64))  # identical fold on every rank


def pipelined_steps(k, issue, collect, exchange=None, depth=2):
    """k steps: `issue()` enqueues this rank's share of one MSM and returns a ticket, `collect(ticket)` waits for it and returns
    the rank's partial sum; at most `depth` shares are in flight.  With an exchange (world > 1) the partial sums are folded
    across ranks, the exchange of one step overlapping the collection of the next.  Returns the results of all k steps in order."""
    results, inflight, pending = [], [], None

    def retire(part):
        nonlocal pending
        if exchange is None:
            results.append(part)
            return
        if pending is not None:
            results.append(exchange.finish(pending))
        pending = exchange.start(part)

    for _ in range(k):
        inflight.append(issue())
        if len(inflight) == depth:
            retire(collect(inflight.pop(0)))
    while inflight:
        retire(collect(inflight.pop(0)))
    if pending is not None:
        results.append(exchange.finish(pending))
    return results
