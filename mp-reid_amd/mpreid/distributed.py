"""One-process-per-GPU evaluation helpers (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).

Partition: the gallery is sharded row-wise across ranks; every rank also encodes a
1/P slice of the queries; the L2-normalised query features are exchanged with ONE all-gather
([nq, 1280] fp32 = 17 MB at Market-1501 scale, ~2 MB per link on the fully connected xGMI mesh);
each rank then owns the [nq, ng_local] column block of the distance matrix, and rank 0 (or the
caller) concatenates the blocks on the host.  No floating-point reduction crosses ranks, so the
result does not depend on the number of ranks.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) slice of range(n) owned by `rank`; sizes differ by at most one"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def all_gather_rows(x: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the row shards of every rank (shard sizes from shard_sizes(n_total, world)).
    Ragged shards are padded to the largest one for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    world = dist.get_world_size()
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    assert x.shape[0] == sizes[dist.get_rank()], (x.shape, sizes)
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    out = torch.empty((world * mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, pad)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def gather_column_blocks_to_host(block: torch.Tensor, dst: int = 0):
    """Host-side concatenation of the per-rank [nq, ng_local] blocks (north_star: "per-shard
    distance blocks concatenated on the host").  Returns the full numpy matrix on `dst`, None elsewhere."""
    import numpy as np
    host = block.cpu().numpy()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return host
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(host, parts, dst=dst)
    return np.concatenate(parts, axis=1) if parts is not None else None
