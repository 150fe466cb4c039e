#!/usr/bin/env python3
"""draw_targets_multiscale on configs[3]'s maps with SPARSE lane sets (1 / 2 polylines of 24 points per frame): lane maps through
the sampler riding in the box-map launch + point splat, or through the one-launch lane raster (lane_raster_multi_kernel)?"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_targets_multiscale, lanes as lanes_mod  # noqa: E402


def gpu_us(fn, n=300, warm=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    B, SH, SW, P = 32, 2160, 3840, 24
    strides = (4.0, 8.0, 16.0)
    g = torch.Generator().manual_seed(7)
    cs, bs = [], []
    for _ in range(B):
        n = int(torch.randint(1, 129, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([SW, SH])
        half = torch.rand(n, 4, generator=g) * 400
        cs.append(c)
        bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
    crb = combine_data(cs, device=dev)
    brb = combine_data(bs, device=dev, other_with_same_sample_sizes=crb)
    maps = [torch.empty((B, int(SH / s), int(SW / s)), device=dev) for s in strides]
    lane_maps = [torch.empty_like(m) for m in maps]
    x0 = torch.rand(B, 8, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, 8, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, 8, P)
    for nl in (1, 2):
        lanes = torch.stack([xs[:, :nl], ys[:, :nl]], -1).contiguous().to(dev)
        best = {True: 1e9, False: 1e9}
        for _ in range(3):
            for rider in (True, False):
                lanes_mod.TARGETS_PREFER_RIDER = rider
                best[rider] = min(best[rider], gpu_us(lambda: draw_targets_multiscale(maps, crb, brb, strides, lane_maps, lanes, 256, 2, clear=True)))
        lanes_mod.TARGETS_PREFER_RIDER = True
        print(json.dumps({"polylines_per_frame": nl, "rider_plus_point_splat_us": round(best[True], 2),
                          "one_launch_lane_raster_us": round(best[False], 2)}), flush=True)


if __name__ == "__main__":
    main()
