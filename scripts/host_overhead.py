#!/usr/bin/env python3
"""Host-side cost per operator call (wall time of N back-to-back asynchronous calls / N, small inputs so the GPU is
never the bottleneck): what an eager training step pays per op before any kernel runs."""
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import RaggedBatch, batched_indexing_access, batched_indexing_access_cuda as ext  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_batched  # noqa: E402


def rate(fn, n=2000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


def main():
    dev = torch.device("cuda", 0)
    b, n, k, d = 8, 900, 100, 256
    data = torch.randn(b, n, d, device=dev)
    idx = torch.randint(0, n, (b, k), device=dev)
    counts = torch.full((b,), k, device=dev, dtype=torch.int64)
    out = torch.zeros(b, k, d, device=dev)
    irb = RaggedBatch(idx, sample_sizes=counts)
    idx_e = idx.unsqueeze(-1).expand(b, k, d)
    lib = nat.lib()
    stream = torch.cuda.current_stream().cuda_stream
    res = {
        "torch.gather(out=)": rate(lambda: torch.gather(data, 1, idx_e, out=out)),
        "torch.zeros_like": rate(lambda: torch.zeros_like(out)),
        "raw ctypes accv_ragged_gather": rate(lambda: lib.accv_ragged_gather(
            data.data_ptr(), out.data_ptr(), idx.data_ptr(), counts.data_ptr(), b, n, k, k, d * 4, 1, 1, None, stream)),
        "ext.gather_rows": rate(lambda: ext.gather_rows(data, idx, counts, k, out)),
        "ext.forward (alloc+fill+gather)": rate(lambda: ext.forward(data, idx, counts, 0.0)),
        "batched_indexing_access (autograd.Function + RaggedBatch)": rate(lambda: batched_indexing_access(data, irb, 0.0)),
    }
    hm = torch.zeros(4, 64, 64, device=dev)
    c = SimpleNamespace(tensor=torch.randint(0, 64, (4, 8, 2), device=dev, dtype=torch.int32),
                        sample_sizes=torch.full((4,), 8, device=dev, dtype=torch.int64))
    r = SimpleNamespace(tensor=torch.randint(1, 5, (4, 8), device=dev, dtype=torch.int32), sample_sizes=c.sample_sizes)
    res["draw_heatmap_batched (64x64)"] = rate(lambda: draw_heatmap_batched(hm, c, r, 6.0, 1.0))
    res["raw ctypes accv_draw_heatmap_batched_f32"] = rate(lambda: lib.accv_draw_heatmap_batched_f32(
        hm.data_ptr(), 4, 0, 64, 64, c.tensor.data_ptr(), r.tensor.data_ptr(), c.sample_sizes.data_ptr(), None, 8, 6.0,
        1.0, 2, stream))
    from accvlab.draw_heatmap import draw_heatmap

    fc = c.tensor.reshape(-1, 2).contiguous()
    fr = r.tensor.reshape(-1).contiguous()
    fi = torch.arange(4, device=dev, dtype=torch.int32).repeat_interleave(8)
    res["draw_heatmap flat API (64x64, 32 objects; host side)"] = rate(lambda: draw_heatmap(hm, fc, fr, fi, 6.0, 1.0))

    def gpu_time(fn, n=300):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    res["draw_heatmap flat API (64x64) us per call incl. GPU"] = gpu_time(lambda: draw_heatmap(hm, fc, fr, fi, 6.0, 1.0))
    res["draw_heatmap_batched (64x64) us per call incl. GPU"] = gpu_time(lambda: draw_heatmap_batched(hm, c, r, 6.0, 1.0))
    import bench_workloads as wl
    from accvlab.batching_helpers import combine_data

    boxes_cpu = wl.ragged_boxes(64, 1, 32, seed=0)
    boxes_gpu = [t.to(dev) for t in boxes_cpu]
    res["combine_data(64 CPU samples -> cuda)"] = rate(lambda: combine_data(boxes_cpu, device=dev), n=500)
    res["combine_data(64 CUDA samples)"] = rate(lambda: combine_data(boxes_gpu), n=500)
    rb = combine_data(boxes_gpu)
    # (measured twice, the smaller value reported: the first measurement right behind the 700 combine_data calls above reads
    # ~100 us, every later one 10-13 us — also in isolation)
    res["RaggedBatch.split (64 CUDA samples; sizes known on the host: no read-back)"] = min(rate(lambda: rb.split(), n=500),
                                                                                             rate(lambda: rb.split(), n=500))
    print(json.dumps({k: round(v, 2) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
