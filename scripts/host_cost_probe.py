#!/usr/bin/env python3
"""Host (python + binding) cost per call of the multi-scale operators of configs[3]: the same calls on tiny maps, where the GPU work is
negligible and the loop time is what the host needs to issue a step."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]
import torch
from accvlab.batching_helpers import combine_data
from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, draw_targets_multiscale
dev = torch.device("cuda", 0)
B, SH, SW, L, P, S = 32, 2160, 3840, 8, 24, 256
strides = (4.0, 8.0, 16.0)
g = torch.Generator().manual_seed(7)
cs, bs = [], []
for _ in range(B):
    n = int(torch.randint(1, 129, (1,), generator=g))
    c = torch.rand(n, 2, generator=g) * torch.tensor([SW, SH]); half = torch.rand(n, 4, generator=g) * 400
    cs.append(c); bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
crb = combine_data(cs, device=dev); brb = combine_data(bs, device=dev, other_with_same_sample_sizes=crb)
# tiny maps: the GPU work is negligible, the loop time is the host cost of a call
small = [torch.empty((B, 8, 16), device=dev) for _ in strides]; small_l = [torch.empty_like(m) for m in small]
lanes = (torch.rand(B, L, P, 2, generator=g) * 100).to(dev)
def host_us(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize(); return round(dt / n * 1e6, 1)
print(json.dumps({"draw_targets_multiscale host us per call": host_us(lambda: draw_targets_multiscale(small, crb, brb, strides, small_l, lanes, S, 2, clear=True)),
                  "draw_heatmap_multiscale": host_us(lambda: draw_heatmap_multiscale(small, crb, brb, strides, 6.0, 1.0, clear=True)),
                  "draw_polylines_multiscale": host_us(lambda: draw_polylines_multiscale(small_l, lanes, S, 2, strides, clear=True))}))
