#!/usr/bin/env python3
"""Per-scale GPU time of the lane raster: the three-launch formulation (sampler -> integer targets -> draw_heatmap_batched with
the small-splat kernel: one-level cull) vs the two-launch path draw_polylines_batched takes for small radii since round 3
(sampler + group boxes -> point splat with its two-level cull), C3's lanes, batch 32."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab.draw_heatmap import draw_polylines_batched  # noqa: E402
from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets  # noqa: E402


def gpu_us(fn, n=200):
    for _ in range(30):
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
    B, SH, SW, L, P = 32, 2160, 3840, 8, 24
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    lanes = torch.stack([xs, ys], -1).to(dev)
    for s in (4, 8, 16):
        hm = torch.empty(B, SH // s, SW // s, device=dev)
        for q in (64, 256):
            one = gpu_us(lambda: _draw_polylines_via_targets(hm, lanes, q, 2, float(s), clear=True))
            two = gpu_us(lambda: draw_polylines_batched(hm, lanes, q, 2, float(s), clear=True))
            print(json.dumps({"stride": s, "samples_per_lane": q, "via_integer_targets_us": round(one, 1), "draw_polylines_batched_us": round(two, 1)}))


if __name__ == "__main__":
    main()
