"""Multi-GPU harness pieces of bench.py that do not need a GPU (so they can be tested with gloo on CPU).

The hot path shards as independent 30 s streams, one model replica per GPU: there is no data-path
collective.  torch.distributed is used for the rendezvous, the barriers around the timed region and the MAX reduction of
the elapsed time only.
"""
from __future__ import annotations

import time


def assign_streams(n_streams: int, world: int):
    """stream s -> rank s % world.  Returns the list of stream lists per rank."""
    out = [[] for _ in range(world)]
    for s in range(n_streams):
        out[s % world].append(s)
    return out


def timed_region(step_fn, steps: int, dist=None, sync_fn=None, device=None):
    """barrier + sync, run `steps` x step_fn, sync + barrier, MAX over ranks.  Returns elapsed seconds (same on all ranks)."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()

    barrier()
    if sync_fn:
        sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    if sync_fn:
        sync_fn()
    barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=device or "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def aggregate(elapsed_s: float, steps: int, world: int):
    """(ms per step of one stream, aggregate ms per chunk over all streams, chunks per second)."""
    ms_per_step = elapsed_s * 1e3 / steps
    return ms_per_step, ms_per_step / world, world * steps / elapsed_s


def broadcast_buffers(dist, torch, views, device="cpu", cap: int = 64, n: int | None = None):
    """One-time weight distribution: broadcast rank 0's buffers into the identically laid out buffers of
    every other rank.  `views` = one flat uint8 tensor per weight buffer, aliasing the buffer (device memory under RCCL,
    host memory under gloo in the CPU test).  Every rank first learns rank 0's buffer count and sizes; if ANY rank's layout
    differs, all ranks skip together (an all-reduce MIN carries the decision, so no rank is left waiting inside a
    collective) and keep the weights they loaded from the model file.  `n` = the rank's true buffer count when it exceeds
    `cap` (then everybody skips too).  Returns bytes broadcast, or None when skipped."""
    n = len(views) if n is None else n
    sizes = [int(v.numel()) for v in views[:cap]]
    meta = torch.tensor([n] + sizes + [0] * (cap - len(sizes)), dtype=torch.int64, device=device)
    ref = meta.clone()
    dist.broadcast(ref, src=0)
    ok = torch.tensor([1 if torch.equal(ref, meta) and 0 < n <= cap else 0], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None
    total = 0
    for v in views:
        dist.broadcast(v, src=0)
        total += int(v.numel())
    return total
