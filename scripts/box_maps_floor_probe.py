#!/usr/bin/env python3
"""Box maps of configs[3] (draw_heatmap_multiscale, strides 4 / 8 / 16 of 3840 x 2160, batch 32, fused clear): the launch with the
config's objects, with no objects at all (every tile stores zeros: the floor of the launch) and with point-like boxes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab.batching_helpers import RaggedBatch, combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_multiscale  # noqa: E402


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
    B, SH, SW = 32, 2160, 3840
    strides = (4.0, 8.0, 16.0)
    g = torch.Generator().manual_seed(7)
    cs, bs, tiny = [], [], []
    for _ in range(B):
        n = int(torch.randint(1, 129, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([SW, SH])
        half = torch.rand(n, 4, generator=g) * 400
        cs.append(c)
        bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
        tiny.append(torch.cat([c - 8.0, c + 8.0], 1))
    crb = combine_data(cs, device=dev)
    brb = combine_data(bs, device=dev, other_with_same_sample_sizes=crb)
    trb = combine_data(tiny, device=dev, other_with_same_sample_sizes=crb)
    none = RaggedBatch(crb.tensor, sample_sizes=torch.zeros_like(crb.sample_sizes))
    none_b = RaggedBatch(brb.tensor, sample_sizes=none.sample_sizes)
    maps = [torch.empty((B, int(SH / s), int(SW / s)), device=dev) for s in strides]
    nbytes = sum(m.numel() * 4 for m in maps)
    row = {}
    for name, c_, b_ in (("config's boxes (half sizes up to 400 px)", crb, brb), ("no objects", none, none_b), ("boxes of 16 px", crb, trb)):
        best = min(gpu_us(lambda: draw_heatmap_multiscale(maps, c_, b_, strides, 6.0, 1.0, clear=True)) for _ in range(3))
        row[name] = {"us": round(best, 2), "frac_of_hbm_peak": round(nbytes / best / 1e3 / 8000.0, 3)}
    print(json.dumps(row))


if __name__ == "__main__":
    main()
