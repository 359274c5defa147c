"""Multi-GPU host logic: one process per GPU, batches shard as contiguous index blocks
(independent units: no data-path collective).  The only exchange step on the path is the
aggregate-verify flag: each rank reduces its flags to one int32 (1 = all valid) and the ranks
MIN-reduce that word (RCCL has sum/prod/min/max but no bit-AND; min over {0,1} is AND).  With
backend "nccl" that is a 4-byte RCCL all-reduce over xGMI; the CPU tests run the same code on gloo.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of [0, n): the first n % world ranks get one extra element."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def collective_device(dist=None):
    """Where the 4-byte flag word lives for the reduce: the GPU for backend nccl (= RCCL over xGMI), the host for gloo."""
    import torch
    if dist is not None and dist.is_initialized() and dist.get_backend() != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def and_reduce_(flag, dist=None):
    """THE exchange step: in-place MIN (= AND over {0,1}) of a one-element int32 tensor over all ranks; returns the tensor
    that holds the result (a host copy when the backend reduces on the host).  No-op for a single rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return flag
    dev = collective_device(dist)
    if flag.device != dev:
        flag = flag.to(dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return flag


def all_valid(local_ok: int, dist=None, device=None) -> int:
    """AND of the per-rank flags.  `dist` is torch.distributed (already initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(bool(local_ok))
    import torch
    t = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device if device is not None else collective_device(dist))
    return int(and_reduce_(t, dist).item())


def max_over_ranks(value: float, dist=None) -> float:
    """Slowest rank's wall time (the bench contract's MAX over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=collective_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_flags(local_flags: np.ndarray, n: int, dist=None) -> np.ndarray:
    """Optional: the full flag vector on every rank (all-gather of the per-shard bytes)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local_flags, dtype=np.uint8)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = max(shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world))
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    mine[: len(local_flags)] = torch.from_numpy(np.ascontiguousarray(local_flags, dtype=np.uint8)).to(dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.zeros(n, dtype=np.uint8)
    for r, part in enumerate(parts):
        lo, hi = shard_bounds(n, r, world)
        out[lo:hi] = part[: hi - lo].cpu().numpy()
    return out


def verify_sharded(verify_fn, pk_xy, msgs, sig_xy, dist=None):
    """Run `verify_fn(pk_shard, msgs_shard, sig_shard) -> uint8 flags` on this rank's block and
    combine: returns (local_flags, (lo, hi), all_valid_over_all_ranks)."""
    n = len(msgs)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(n, rank, world)
    flags = np.asarray(verify_fn(pk_xy[lo:hi], msgs[lo:hi], sig_xy[lo:hi]), dtype=np.uint8) if hi > lo else np.zeros(0, np.uint8)
    return flags, (lo, hi), all_valid(int(flags.all()), dist)
