"""Multi-GPU plumbing: frames and files are independent, so ranks share nothing until the final gather.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  The only
collective on the data path is the gather of rendered RGBA tiles to one rank.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Frame f belongs to rank f mod world."""
    return list(range(rank, n_frames, world))


def balance_files(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Greedy bin-packing of files (by byte size, largest first) onto ranks; returns file indices per rank."""
    bins: List[List[int]] = [[] for _ in range(world)]
    load = [0] * world
    for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        bins[r].append(i)
        load[r] += sizes[i]
    return [sorted(b) for b in bins]


def gather_tiles(tiles: torch.Tensor, dst: int = 0, out: Optional[List[torch.Tensor]] = None):
    """Every rank contributes `tiles` [F, H, W, 4] uint8 (same shape on all ranks); rank `dst` receives the list
    of all ranks' tiles (pre-allocated `out` is reused), other ranks get None."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if rank == dst:
        if out is None:
            out = [torch.empty_like(tiles) for _ in range(world)]
        dist.gather(tiles, out, dst=dst)
        return out
    dist.gather(tiles, None, dst=dst)
    return None


def interleave_gathered(per_rank: List[torch.Tensor], n_frames: int) -> torch.Tensor:
    """Undo shard_frames: per_rank[r][i] is frame r + i*world; returns tiles in frame order [n_frames, ...]."""
    world = len(per_rank)
    out = torch.empty((n_frames,) + tuple(per_rank[0].shape[1:]), dtype=per_rank[0].dtype, device=per_rank[0].device)
    for r, t in enumerate(per_rank):
        idx = list(range(r, n_frames, world))
        out[idx] = t[: len(idx)]
    return out
