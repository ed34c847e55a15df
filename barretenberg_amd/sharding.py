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
    """All-gather of the ranks' 96-byte partial sums + the identical host fold.  Everything a step needs is allocated once: a ring of
    pinned host / device send buffers and one flat device / pinned host receive buffer per slot, so a step costs one small H2D copy, one
    `all_gather_into_tensor` and one D2H copy -- a rank's share of an 8-way split is ~0.2 ms of GPU time, and a per-step `torch.empty` x N,
    a pageable `.to(device)` and a `torch.stack(...).cpu()` (the first version) are of the same order on the host thread."""
    RING = 4  # exchanges in flight at most (pipelined_steps keeps one pending)

    def __init__(self, lib, world, device):
        """lib: BbGpu (only g1_sum is used: host arithmetic); device: where the 96-byte tensors live (cuda for nccl, cpu for gloo)"""
        self.lib, self.world, self.device = lib, world, device
        self.on_gpu = device.type == "cuda"
        host = dict(dtype=torch.int64, pin_memory=self.on_gpu)
        self.send_h = [torch.empty(12, **host) for _ in range(self.RING)]
        self.recv_h = [torch.empty(world * 12, **host) for _ in range(self.RING)]
        if self.on_gpu:
            self.send_d = [torch.empty(12, dtype=torch.int64, device=device) for _ in range(self.RING)]
            self.recv_d = [torch.empty(world * 12, dtype=torch.int64, device=device) for _ in range(self.RING)]
            self.done = [torch.cuda.Event() for _ in range(self.RING)]
        else:
            self.send_d, self.recv_d = self.send_h, self.recv_h
        # flat (all_gather_into_tensor) or list form: settled ONCE here, by a collective every rank runs in the same order, and agreed by a
        # MIN all-reduce -- never per step, where one rank falling back alone would pair a flat call with a list call
        self.flat = self._settle_flat()
        self.count = 0

    def _settle_flat(self):
        ok = 0
        if hasattr(dist, "all_gather_into_tensor"):
            try:
                work = dist.all_gather_into_tensor(self.recv_d[0], self.send_d[0].zero_(), async_op=True)
                work.wait()
                if self.on_gpu:
                    torch.cuda.synchronize(self.device)
                ok = 1
            except (RuntimeError, NotImplementedError):  # a backend without the flat form raises before it communicates, on every rank alike
                ok = 0
        t = torch.tensor([ok], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def start(self, part):
        k = self.count % self.RING
        self.count += 1
        self.send_h[k].numpy()[:] = np.ascontiguousarray(part, dtype=np.uint64).view(np.int64)
        if self.on_gpu:
            self.send_d[k].copy_(self.send_h[k], non_blocking=True)
        if self.flat:
            work = dist.all_gather_into_tensor(self.recv_d[k], self.send_d[k], async_op=True)
        else:
            work = dist.all_gather(list(self.recv_d[k].view(self.world, 12).unbind(0)), self.send_d[k], async_op=True)
        return work, k

    def finish(self, handle):
        work, k = handle
        work.wait()  # nccl: the current stream waits for the collective; gloo: the host does
        if self.on_gpu:
            self.recv_h[k].copy_(self.recv_d[k], non_blocking=True)
            self.done[k].record()
            self.done[k].synchronize()
        return self.lib.g1_sum(self.recv_h[k].numpy().view(np.uint64).reshape(self.world, 12).copy())  # identical fold on every rank


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
