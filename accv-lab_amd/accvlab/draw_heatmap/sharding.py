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


# ------------------------------------------------------------------------------------------------------------------
# Control flow of a frame-sharded, one-process-per-GPU job.  bench.py is built from these functions, and the
# world-size-2 gloo test (tests/test_sharding_gloo.py) runs the SAME functions on CPU.
def rank_layout(env=None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) as set by ``python -m torch.distributed.run`` (1-process defaults otherwise)."""
    import os

    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def rank_seed(base_seed: int, rank: int) -> int:
    """Every rank draws its OWN frames (weak scaling): seed = base + rank."""
    return int(base_seed) + int(rank)


def init_process_group(world: int, backend: str = "nccl", device=None):
    """``torch.distributed`` module with an initialised default group, or None for a single process.  ``nccl`` is
    RCCL on ROCm (one rank per GPU); ``gloo`` lets several ranks share a device or run on CPU (tests, rehearsals)."""
    if world <= 1:
        return None
    import torch.distributed as dist

    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return dist


def job_barrier(dist, sync=None) -> None:
    """device sync + barrier + device sync: the bracket of a timed region (bench contract)."""
    if sync is not None:
        sync()
    if dist is not None:
        dist.barrier()
    if sync is not None:
        sync()


def timed_steps(step, steps: int, dist=None, sync=None, clock=None) -> float:
    """Run ``step()`` exactly ``steps`` times between two ``job_barrier`` brackets; wall milliseconds per step of THIS
    rank (reduce with :func:`max_over_ranks`)."""
    import time

    clock = clock or time.perf_counter
    job_barrier(dist, sync)
    t0 = clock()
    for _ in range(steps):
        step()
    job_barrier(dist, sync)
    return (clock() - t0) * 1e3 / max(1, steps)


def job_throughput(units_per_rank: int, world: int, ms_per_step_max: float) -> float:
    """Whole-job units per second: every rank processed ``units_per_rank`` per step, the step took the slowest
    rank's time."""
    return world * units_per_rank / (ms_per_step_max * 1e-3)


# ------------------------------------------------------------------------------------------------------------------
# Strong scaling: ONE batch of `total` frames is cut contiguously over the ranks
# (64 -> 64 / 32 / 16 / 8 frames per GPU at 1 / 2 / 4 / 8 GPUs); every rank draws only its slice.
def strong_scaling_run(make_step, total: int, rank: int, world: int, steps: int, warmup: int, dist=None, sync=None,
                       device=None) -> dict:
    """Time ``steps`` draws of this rank's :func:`shard_range` slice of a ``total``-frame batch between the same barrier
    brackets as the weak mode and reduce with MAX over ranks.

    ``make_step(begin, end)`` returns the callable that draws frames ``[begin, end)`` (called once, outside the timed
    region); a rank with an empty slice (``world > total``) only takes part in the barriers.  Returns the per-rank and
    whole-job figures; ``frames_per_s`` = ``total`` / slowest rank's time per step.
    """
    begin, end = shard_range(total, rank, world)
    step = make_step(begin, end) if end > begin else (lambda: None)
    for _ in range(max(0, int(warmup))):
        step()
    ms = timed_steps(step, steps, dist=dist, sync=sync)
    ms_max = max_over_ranks(ms, device=device)
    return {"total_frames": int(total), "frames_this_rank": end - begin, "range_this_rank": [begin, end],
            "frames_per_rank": [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)],
            "ms_this_rank": ms, "ms_per_step": ms_max, "frames_per_s": total / (ms_max * 1e-3), "steps": int(steps)}
