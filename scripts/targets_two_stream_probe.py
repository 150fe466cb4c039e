#!/usr/bin/env python3
"""configs[3]: would running the box maps and the point splat CONCURRENTLY (two streams, both behind the sampler's own launch)
beat draw_targets_multiscale (sampler riding in the box-map launch, point splat behind it on one stream)?  Both kernels are bound
by the HBM store stream, so the overlap can only hide their launch boundaries and tails."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, draw_targets_multiscale, sample_lanes  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, SH, SW, L, P, S = 32, 2160, 3840, 8, 24, 256
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
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    lanes = torch.stack([xs, ys], -1).to(dev)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    work = torch.empty(nat.lib().accv_draw_points_workspace_bytes(B, L * S), dtype=torch.uint8, device=dev)

    def one_stream():
        draw_targets_multiscale(maps, crb, brb, strides, lane_maps, lanes, S, 2, clear=True)

    def two_streams():
        samples = sample_lanes(lanes, S, group_boxes_ptr=work.data_ptr())
        ev = torch.cuda.Event()
        ev.record(main_s)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=True)
            done = torch.cuda.Event()
            done.record(side)
        draw_polylines_multiscale(lane_maps, lanes, S, 2, strides, clear=True, _presampled=(samples, work))
        main_s.wait_event(done)

    def wall_us(fn, n=500, warm=200):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    best = {"one stream (draw_targets_multiscale)": 1e9, "sampler, then box maps || point splat on two streams": 1e9}
    for _ in range(3):
        best["one stream (draw_targets_multiscale)"] = min(best["one stream (draw_targets_multiscale)"], wall_us(one_stream))
        best["sampler, then box maps || point splat on two streams"] = min(best["sampler, then box maps || point splat on two streams"], wall_us(two_streams))
    print(json.dumps({k: round(v, 2) for k, v in best.items()}))


if __name__ == "__main__":
    main()
