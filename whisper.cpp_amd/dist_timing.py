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


def share_bytes(dist, torch, payload: bytes | None, n: int, device="cpu") -> bytes:
    """rank 0's `n` bytes on every rank (the 128-byte RCCL unique id of the plugin's weight broadcast travels this way, over the
    harness's own process group)."""
    t = torch.tensor(list(payload) if payload is not None else [0] * n, dtype=torch.uint8, device=device)
    assert t.numel() == n
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def all_ranks_ok(dist, torch, ok_local: bool, device="cpu") -> bool:
    """True only if EVERY rank reports success (all-reduce MIN): a replica whose weights could not be verified makes the whole job
    fail together instead of benchmarking a different model on one GPU."""
    ok = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return int(ok.item()) == 1
