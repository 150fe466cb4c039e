#!/usr/bin/env python3
"""rocprofv3 target: the lane raster of a sparse scene (2 polylines x 24 points per frame, 256 samples, radius 2, configs[3]'s
maps, batch 32) 300 times through the fused kernel and 300 times through sampler + point splat."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab.draw_heatmap import draw_polylines_multiscale, lanes as lanes_mod  # noqa: E402

dev = torch.device("cuda", 0)
B, SH, SW, L, P = 32, 2160, 3840, 2, 24
g = torch.Generator().manual_seed(7)
x0 = torch.rand(B, 8, 1, generator=g) * SW
t_ = torch.linspace(0, 1, P).view(1, 1, P)
xs = x0 + (torch.rand(B, 8, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
ys = SH * (1 - 0.9 * t_).expand(B, 8, P)
lanes = torch.stack([xs[:, :L], ys[:, :L]], -1).contiguous().to(dev)
strides = (4.0, 8.0, 16.0)
maps = [torch.empty(B, int(SH / s), int(SW / s), device=dev) for s in strides]
for fused in (True, False):
    lanes_mod.FUSED_SAMPLER = fused
    for _ in range(300):
        draw_polylines_multiscale(maps, lanes, 256, 2, strides, clear=True)
    torch.cuda.synchronize()
lanes_mod.FUSED_SAMPLER = True
print("ok")
