"""N>1 path on CPU: two gloo ranks shard a batch of frames, "draw" their slice with the CPU oracle (test
infrastructure standing in for the GPU op), and the optional all-gather must reproduce the single-process result
frame for frame.  Also covers max-over-ranks and uneven shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench_workloads as wl
from accvlab.draw_heatmap.sharding import (all_gather_heatmaps, init_process_group, job_throughput, max_over_ranks,
                                           rank_layout, rank_seed, shard_range, strong_scaling_run, timed_steps)
from oracle import h1 as oracle

H, W, TOTAL = 40, 64, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _draw(centers_l, radii_l, lo, hi):
    cpad, sizes = wl.pad_ragged(centers_l[lo:hi])
    rpad, _ = wl.pad_ragged(radii_l[lo:hi])
    hm = np.zeros((hi - lo, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(hm, cpad.numpy(), rpad.numpy(), sizes.numpy(), clear=True)
    return torch.from_numpy(hm)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        centers_l, radii_l = wl.heatmap_objects(TOTAL, H, W, 0, 6, "A", seed=3)
        lo, hi = shard_range(TOTAL, rank, world)
        local = _draw(centers_l, radii_l, lo, hi)
        full = all_gather_heatmaps(local, total_frames=TOTAL)
        slow = max_over_ranks(1.0 + rank)
        torch.save({"full": full, "slow": slow, "range": (lo, hi)}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_shard_range_properties():
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            lens = [e - b for b, e in spans]
            assert max(lens) - min(lens) <= 1
    assert [shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_gloo_sharded_draw_and_gather(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    centers_l, radii_l = wl.heatmap_objects(TOTAL, H, W, 0, 6, "A", seed=3)
    ref = _draw(centers_l, radii_l, 0, TOTAL)
    outs = [torch.load(os.path.join(str(tmp_path), f"r{r}.pt")) for r in range(world)]
    assert [o["range"] for o in outs] == [(0, 3), (3, 5)]          # uneven shards exercise the padded gather
    for o in outs:
        assert torch.equal(o["full"], ref)
        assert o["slow"] == 2.0


# ---- the control flow of bench.py (rank layout from the launcher's environment, per-rank seeds, barrier-bracketed timed
# region, MAX over ranks, whole-job throughput) — the very functions bench.py calls, on two gloo ranks
def _bench_worker(rank, world, port, out_dir):
    import time

    env = {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world)}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    r, lr, w = rank_layout(env)
    d = init_process_group(w, backend="gloo")
    try:
        seed = rank_seed(42, r)
        centers_l, radii_l = wl.heatmap_objects(4, H, W, 1, 6, "A", seed=seed)
        n_objects = sum(int(x.shape[0]) for x in radii_l)
        checksum = int(sum(int(c.to(torch.int64).sum()) * (i + 1) for i, c in enumerate(centers_l)))
        calls = []

        def step():                       # rank 1 is the slow one
            calls.append(1)
            time.sleep(0.002 * (1 + r))

        ms = timed_steps(step, 5, dist=d, sync=None)
        ms_max = max_over_ranks(ms)
        torch.save({"rank": (r, lr, w), "seed": seed, "n_objects": n_objects, "checksum": checksum, "calls": len(calls), "ms": ms, "ms_max": ms_max,
                    "value": job_throughput(4, w, ms_max)}, os.path.join(out_dir, f"b{rank}.pt"))
        d.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_control_flow_on_two_gloo_ranks(tmp_path):
    world = 2
    mp.start_processes(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    outs = [torch.load(os.path.join(str(tmp_path), f"b{r}.pt")) for r in range(world)]
    assert [o["rank"] for o in outs] == [(0, 0, 2), (1, 1, 2)]
    assert [o["seed"] for o in outs] == [42, 43]                        # every rank draws its own frames
    assert outs[0]["checksum"] != outs[1]["checksum"]                   # ... the DATA differ, not only the seed integers
    assert all(o["calls"] == 5 for o in outs)                           # exactly K steps inside the timed region
    assert outs[0]["ms_max"] == outs[1]["ms_max"] >= outs[1]["ms"] - 1e-9   # MAX over ranks, identical on both
    assert outs[1]["ms"] >= 4.0 and outs[0]["ms_max"] >= 4.0            # the slow rank (2 x 2 ms sleeps) sets the time
    # the barrier keeps the fast rank inside the region until the slow one is done
    assert outs[0]["ms"] >= 0.9 * outs[1]["ms"]
    assert abs(outs[0]["value"] - 2 * 4 / (outs[0]["ms_max"] * 1e-3)) < 1e-6
    assert rank_layout({}) == (0, 0, 1) and init_process_group(1) is None


# ---- strong scaling: ONE batch cut over the ranks with shard_range; the function bench.py calls
S_TOTAL = 7


def _strong_worker(rank, world, port, out_dir):
    import time

    env = {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world)}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **env)
    r, _, w = rank_layout(env)
    d = init_process_group(w, backend="gloo")
    try:
        centers_l, radii_l = wl.heatmap_objects(S_TOTAL, H, W, 1, 6, "A", seed=42)     # the SAME batch on every rank
        drawn, made = [], []

        def make_step(lo, hi):
            made.append((lo, hi))

            def step():
                drawn.append(_draw(centers_l, radii_l, lo, hi))
                time.sleep(0.001 * (hi - lo))          # cost grows with the slice: rank 0 (4 frames) is the slow one
            return step

        res = strong_scaling_run(make_step, S_TOTAL, r, w, steps=3, warmup=2, dist=d, sync=None)
        res.update(made=made, calls=len(drawn), last=drawn[-1])
        torch.save(res, os.path.join(out_dir, f"s{rank}.pt"))
        d.barrier()
    finally:
        dist.destroy_process_group()


def test_strong_scaling_split_on_two_gloo_ranks(tmp_path):
    world = 2
    mp.start_processes(_strong_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    outs = [torch.load(os.path.join(str(tmp_path), f"s{r}.pt")) for r in range(world)]
    assert [o["range_this_rank"] for o in outs] == [[0, 4], [4, 7]] and [o["made"] for o in outs] == [[(0, 4)], [(4, 7)]]
    assert all(o["frames_per_rank"] == [4, 3] and o["total_frames"] == S_TOTAL for o in outs)
    assert all(o["calls"] == 2 + 3 for o in outs)                        # warm-up + exactly K timed steps
    assert outs[0]["ms_per_step"] == outs[1]["ms_per_step"] >= 4.0      # MAX over ranks: the 4-frame slice sets the time
    assert outs[0]["frames_per_s"] == outs[1]["frames_per_s"] == S_TOTAL / (outs[0]["ms_per_step"] * 1e-3)
    # the slices put side by side ARE the single-process batch
    centers_l, radii_l = wl.heatmap_objects(S_TOTAL, H, W, 1, 6, "A", seed=42)
    assert torch.equal(torch.cat([o["last"] for o in outs]), _draw(centers_l, radii_l, 0, S_TOTAL))


def test_strong_scaling_single_process_and_empty_slices():
    seen = []
    res = strong_scaling_run(lambda lo, hi: (lambda: seen.append((lo, hi))), 5, 0, 1, steps=4, warmup=1)
    assert seen == [(0, 5)] * 5 and res["frames_per_rank"] == [5] and res["frames_per_s"] > 0
    # more ranks than frames: the surplus rank draws nothing (make_step is never called for it) but still reports
    res = strong_scaling_run(lambda lo, hi: pytest.fail("empty slice must not build a step"), 2, 2, 3, steps=1, warmup=0)
    assert res["frames_this_rank"] == 0 and res["frames_per_rank"] == [1, 1, 0]
