#!/usr/bin/env python3
"""Kernel-time drift over a long back-to-back run (blocks of 100 launches timed with HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]
import torch
import bench_workloads as wl
from accvlab.batching_helpers import combine_data
from accvlab.draw_heatmap import draw_heatmap_batched

dev = torch.device("cuda", 0)
B, H, W = 64, 1080, 1920
cl, rl = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=42)
c = combine_data(cl, device=dev); r = combine_data(rl, device=dev, other_with_same_sample_sizes=c)
hm = torch.empty((B, H, W), device=dev)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
torch.cuda.synchronize()
evs[0].record()
for b in range(blocks):
    for _ in range(100):
        draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=True)
    evs[b + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) / 100 for i in range(blocks)]
print("us per launch by block of 100:", " ".join(f"{t*1e3:.1f}" for t in ts))
