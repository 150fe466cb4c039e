#!/usr/bin/env python3
"""Config 3 (box maps + lane raster of a 32-frame 3840x2160 batch): does running the polyline sampler on a side stream,
under the box-map launch, pay?  Sequential = the three launches of draw_heatmap_multiscale + draw_polylines_multiscale on one
stream; overlapped = sampler (with group boxes) on a side stream while the box maps are drawn, then the lane splat."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, sample_lanes  # noqa: E402


def gpu_us(fn, n=300, warm=100):
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
    lib = nat.lib()
    B, SH, SW, L, P, Q = 32, 2160, 3840, 8, 24, 256
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
    n = L * Q
    ws_bytes = lib.accv_draw_points_workspace_bytes(B, n)
    work = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.full((B,), n, dtype=torch.int32, device=dev)
    k = len(maps)
    ptrs = (ctypes.c_void_p * k)(*[m.data_ptr() for m in lane_maps])
    hs = (ctypes.c_int * k)(*[m.size(1) for m in lane_maps])
    ws_ = (ctypes.c_int * k)(*[m.size(2) for m in lane_maps])
    st = (ctypes.c_float * k)(*strides)
    side = torch.cuda.Stream()
    keep = {}

    def sequential():
        draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=True)
        draw_polylines_multiscale(lane_maps, lanes, Q, 2, strides, clear=True)

    def overlapped():
        main_s = torch.cuda.current_stream()
        side.wait_event(main_s.record_event())
        with torch.cuda.stream(side):
            keep["samples"] = sample_lanes(lanes, Q, group_boxes_ptr=work.data_ptr())
            done = side.record_event()
        draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=True)
        main_s.wait_event(done)
        nat.check(lib.accv_draw_points_multiscale_f32(ptrs, hs, ws_, st, k, B, keep["samples"].data_ptr(), counts.data_ptr(), n, 2,
                                                      6.0, 1.0, nat.HM_CLEAR | nat.HM_GROUP_BOXES_GIVEN, work.data_ptr(), ws_bytes,
                                                      main_s.cuda_stream), "points")

    def lanes_first():      # same stream, lane launches first: the box kernel hides nothing, only the order changes
        draw_polylines_multiscale(lane_maps, lanes, Q, 2, strides, clear=True)
        draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=True)

    out = {}
    for _ in range(3):
        for name, fn in (("sequential", sequential), ("sampler on a side stream", overlapped), ("lanes first", lanes_first)):
            out.setdefault(name, []).append(round(gpu_us(fn), 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
