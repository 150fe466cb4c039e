"""The RCCL calls the sharded job makes (tests/test_sharding_gloo.py covers the control flow on two gloo ranks; N > 1 GPUs
are the driver's to launch): a one-rank "nccl" group on cuda:0 must initialise exactly the way bench.py initialises it
(device_id given), and barrier / MAX all-reduce of the step time / all-gather of a shard must run on it."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_nccl_group_runs_the_job_collectives():
    import torch.distributed as dist

    from accvlab.draw_heatmap.sharding import all_gather_heatmaps, job_barrier, max_over_ranks, timed_steps

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        job_barrier(dist, torch.cuda.synchronize)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == 1.25
        shard = torch.arange(2 * 3 * 4, dtype=torch.float32, device=dev).reshape(2, 3, 4)
        out = torch.empty_like(shard)
        dist.all_gather_into_tensor(out, shard)
        assert torch.equal(out, shard)
        # the helpers short-circuit for one rank; they must still accept the initialised group
        assert max_over_ranks(2.5, device=dev) == 2.5
        assert torch.equal(all_gather_heatmaps(shard, total_frames=2), shard)
        ms = timed_steps(lambda: shard.add_(0.0), 3, dist, torch.cuda.synchronize)
        assert ms >= 0.0
    finally:
        dist.destroy_process_group()
