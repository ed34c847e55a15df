"""One MSM split over ranks: the one exchange step and the pipelined step loop bench.py runs.

Rank g computes the partial sum of its share (a point range, a window range or a row range of the same MSM:
bbgpu_msm_g1_device_async and relatives); the partial sums (96 bytes per rank, normalised) are all-gathered -- RCCL over xGMI on
the GPUs, gloo in the CPU tests; RCCL has no G1 reduction operator, so "all-reduce" = all-gather + the identical host fold
bbgpu_g1_sum on every rank (the reference adds its threads' partial sums the same way, scalar_multiplication.cpp:755-761).

What a rank's host thread pays per step matters here: at N = 8 a share is ~0.18 ms of GPU time.  Two things keep the host below it:
  * the partial sums of the `group` shares in flight travel in ONE all-gather of group x 96 bytes (every MSM is still folded from its
    own N partial sums): one H2D copy, one collective, one D2H copy and one event per `group` steps instead of per step;
  * the shares are ISSUED by a helper thread (issuer=True) while the calling thread collects and exchanges: the ~15 kernel launches of
    an issue and the wait + fold of a collect then overlap (the library's waits block outside its mutex for exactly this).
"""
import queue
import threading
import time

import numpy as np
import torch
import torch.distributed as dist


class PartialSumExchange:
    """All-gather of the ranks' 96-byte partial sums of up to `group` MSMs at a time + the identical host fold.  Everything a step needs is
    allocated once: a ring of pinned host / device send buffers and one flat device / pinned host receive buffer per slot."""
    RING = 4  # exchanges in flight at most (pipelined_steps keeps one pending)

    def __init__(self, lib, world, device, group=1):
        """lib: BbGpu (only g1_sum is used: host arithmetic); device: where the partial sums live (cuda for nccl, cpu for gloo);
        group: MSMs whose partial sums share one collective (a shorter tail group is padded with points at infinity)"""
        self.lib, self.world, self.device, self.group = lib, world, device, max(1, int(group))
        self.on_gpu = device.type == "cuda"
        words = 12 * self.group
        host = dict(dtype=torch.int64, pin_memory=self.on_gpu)
        self.send_h = [torch.empty(words, **host) for _ in range(self.RING)]
        self.recv_h = [torch.empty(world * words, **host) for _ in range(self.RING)]
        if self.on_gpu:
            self.send_d = [torch.empty(words, dtype=torch.int64, device=device) for _ in range(self.RING)]
            self.recv_d = [torch.empty(world * words, dtype=torch.int64, device=device) for _ in range(self.RING)]
            self.done = [torch.cuda.Event() for _ in range(self.RING)]
        else:
            self.send_d, self.recv_d = self.send_h, self.recv_h
        self.infinity = np.zeros(12, dtype=np.uint64)
        self.infinity[7] = np.uint64(1 << 63)
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

    def start(self, parts):
        """parts: this rank's partial sum of ONE MSM (12 limbs) or a list of up to `group` of them; returns a handle for finish()"""
        if isinstance(parts, np.ndarray) and parts.ndim == 1:
            parts = [parts]
        m = len(parts)
        assert 1 <= m <= self.group
        k = self.count % self.RING
        self.count += 1
        buf = self.send_h[k].numpy().view(np.uint64).reshape(self.group, 12)
        for j in range(self.group):
            buf[j] = parts[j] if j < m else self.infinity
        if self.on_gpu:
            self.send_d[k].copy_(self.send_h[k], non_blocking=True)
        if self.flat:
            work = dist.all_gather_into_tensor(self.recv_d[k], self.send_d[k], async_op=True)
        else:
            work = dist.all_gather(list(self.recv_d[k].view(self.world, 12 * self.group).unbind(0)), self.send_d[k], async_op=True)
        return work, k, m

    def finish(self, handle):
        """the folded results of the handle's MSMs: one point for a single-part start(), else a list in the order given to start()"""
        work, k, m = handle
        work.wait()  # nccl: the current stream waits for the collective; gloo: the host does
        if self.on_gpu:
            self.recv_h[k].copy_(self.recv_d[k], non_blocking=True)
            self.done[k].record()
            self.done[k].synchronize()
        got = self.recv_h[k].numpy().view(np.uint64).reshape(self.world, self.group, 12)
        out = [self.lib.g1_sum(np.ascontiguousarray(got[:, j, :])) for j in range(m)]  # identical fold on every rank
        return out[0] if (m == 1 and self.group == 1) else out


class StepClock:
    """host time a rank's thread(s) spend per step in the four things a step is made of (perf_counter around the calls; ~0.1 us each)"""
    KEYS = ("issue", "wait", "exchange_start", "exchange_finish")

    def __init__(self):
        self.t = dict.fromkeys(self.KEYS, 0.0)
        self.steps = 0

    def add(self, key, dt):
        self.t[key] += dt

    def per_step_us(self):
        return {k: (v / self.steps * 1e6 if self.steps else 0.0) for k, v in self.t.items()}


def pipelined_steps(k, issue, collect, exchange=None, depth=2, clock=None, issuer=False, issuer_device=None):
    """k steps: `issue()` enqueues this rank's share of one MSM and returns a ticket, `collect(ticket)` waits for it and returns
    the rank's partial sum; at most `depth` shares are in flight.  With an exchange (world > 1) the partial sums are folded across
    ranks, `exchange.group` of them per collective, the exchange of one group overlapping the collection of the next.
    issuer=True: the issue() calls run on a helper thread (bounded to `depth` ahead of the collector), the calling thread collects and
    exchanges; issuer_device = the rank's GPU index (the current device is per thread: a fresh thread starts on device 0).
    Returns the results of all k steps in order."""
    results, pending, batch = [], None, []
    group = exchange.group if exchange is not None else 1
    now = time.perf_counter

    def flush():
        nonlocal pending, batch
        if not batch:
            return
        t0 = now()
        if pending is not None:
            got = exchange.finish(pending)
            results.extend(got if isinstance(got, list) else [got])
        t1 = now()
        pending = exchange.start(batch if group > 1 else batch[0])
        t2 = now()
        if clock is not None:
            clock.add("exchange_finish", t1 - t0)
            clock.add("exchange_start", t2 - t1)
        batch = []

    def retire(part):
        if exchange is None:
            results.append(part)
            return
        batch.append(part)
        if len(batch) == group:
            flush()

    def timed_issue():
        t0 = now()
        tk = issue()
        if clock is not None:
            clock.add("issue", now() - t0)
        return tk

    def timed_collect(tk):
        t0 = now()
        part = collect(tk)
        if clock is not None:
            clock.add("wait", now() - t0)
        return part

    def abandon(tickets):
        """a step failed: nothing of this call may stay in flight -- the tickets already issued are collected (their MSM slots would stay pending
        otherwise), the exchange already started is finished (the peers are inside that collective); the caller then re-raises, and a rank that
        dies non-zero is what ends the others (bench.py's launcher), since they would wait in the NEXT collective for ever"""
        nonlocal pending
        for tk in tickets:
            try:
                collect(tk)
            except Exception:  # noqa: BLE001 -- the first failure is the one reported
                pass
        if pending is not None:
            try:
                exchange.finish(pending)
            except Exception:  # noqa: BLE001
                pass
            pending = None

    if issuer and k > 0:
        q, room, err, stop = queue.Queue(), threading.Semaphore(depth), [], object()  # `stop`: a ticket may legitimately be None (an empty share)

        def run():
            try:
                if issuer_device is not None and torch.cuda.is_available():
                    torch.cuda.set_device(issuer_device)
                for _ in range(k):
                    room.acquire()
                    q.put(timed_issue())
            except BaseException as exc:  # noqa: BLE001 -- handed to the collecting thread
                err.append(exc)
            q.put(stop)

        th = threading.Thread(target=run, name="bbgpu-issuer", daemon=True)
        th.start()
        try:
            while True:
                tk = q.get()  # in issue order: every ticket issued before a failure comes out before the marker
                if tk is stop:
                    break
                try:
                    part = timed_collect(tk)
                finally:
                    room.release()
                retire(part)
        except BaseException:
            th.join()
            left = []
            while not q.empty():
                tk = q.get()
                if tk is not stop:
                    left.append(tk)
            abandon(left)
            raise
        th.join()
        if err:
            abandon([])
            raise err[0]
    else:
        inflight = []
        try:
            for _ in range(k):
                inflight.append(timed_issue())
                if len(inflight) == depth:
                    retire(timed_collect(inflight.pop(0)))
            while inflight:
                retire(timed_collect(inflight.pop(0)))
        except BaseException:
            abandon(inflight)
            raise
    if exchange is not None:
        flush()
        if pending is not None:
            t0 = now()
            got = exchange.finish(pending)
            results.extend(got if isinstance(got, list) else [got])
            if clock is not None:
                clock.add("exchange_finish", now() - t0)
    if clock is not None:
        clock.steps += k
    return results
