"""Frame sharding for multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The heat-map path shards embarrassingly: every (sample, plane) is independent, so ranks split the batch dimension
contiguously and run the single-GPU op on their slice — no data-path collective.  The only collective offered is
an OPTIONAL all-gather of the finished maps for consumers that want the whole batch on every GPU (at 1920x1080 that
moves 8.3 MB per frame per peer over xGMI and costs several times the draw itself, so it is off by default;
SURVEY.md §8e).  The reference has no multi-GPU code for this path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of `total` frames owned by `rank`; sizes differ by at most one, earlier ranks get the
    larger shards."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"invalid rank {rank} for world size {world}")
    base, extra = divmod(int(total), world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX-reduce a python float over the group (used for the step time of a sharded run)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0])


def all_gather_heatmaps(local: torch.Tensor, total_frames: Optional[int] = None, group=None) -> torch.Tensor:
    """Gather the per-rank heat-map shards ``[b_r, ...]`` into ``[sum b_r, ...]`` on every rank (rank order).

    Equal shards use one ``all_gather_into_tensor`` (a single RCCL collective on GPUs); unequal shards are padded to
    the largest shard first.  ``total_frames`` (if given) is checked against the result.
    """
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    local = local.contiguous()
    sizes = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    counts = [int(s[0]) for s in all_sizes]
    largest = max(counts)
    if all(c == largest for c in counts):
        out = torch.empty((world * largest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        padded = torch.zeros((largest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        buf = torch.empty((world * largest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        out = torch.cat([buf[r * largest: r * largest + c] for r, c in enumerate(counts)], dim=0)
    if total_frames is not None and out.shape[0] != total_frames:
        raise RuntimeError(f"gathered {out.shape[0]} frames, expected {total_frames}")
    return out
