#!/usr/bin/env python3
"""A/B of the scale order inside the multi-scale launches (BASELINE config 3 shapes): coarse scales first (default) against
the caller's order (ACCV_HM_CALLER_SCALE_ORDER), for the box maps and the lane raster.  Interleaved rounds, GPU events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale  # noqa: E402
from accvlab.draw_heatmap import ops  # noqa: E402


def gpu_us(fn, n=300, warm=50):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    B, SH, SW = 32, 2160, 3840
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
    L, P = 8, 24
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    lanes = torch.stack([xs, ys], -1).to(dev)
    cases = {
        "box maps (draw_heatmap_multiscale, clear)": lambda: draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=True),
        "box maps (in-place)": lambda: draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0),
        "lane raster (draw_polylines_multiscale, clear)": lambda: draw_polylines_multiscale(maps, lanes, 256, 2, strides, clear=True),
    }
    res = {k: {"coarse first": [], "caller order": []} for k in cases}
    for _ in range(3):
        for name, fn in cases.items():
            for label, flag in (("coarse first", 0), ("caller order", nat.HM_CALLER_SCALE_ORDER)):
                ops._FORCED_FLAGS = flag
                res[name][label].append(round(gpu_us(fn), 2))
    ops._FORCED_FLAGS = 0
    for name, r in res.items():
        print(json.dumps({"case": name, **{k: {"us": v, "median": sorted(v)[1]} for k, v in r.items()}}))


if __name__ == "__main__":
    main()
