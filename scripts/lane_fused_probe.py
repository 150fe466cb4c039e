#!/usr/bin/env python3
"""Lane raster on configs[3]'s maps (strides 4 / 8 / 16 of 3840 x 2160, batch 32, 256 samples, radius 2): the fused kernel (tile
waves sample the polylines themselves, one launch; frames of at most 64 point slots) against sampler + point splat (two
launches), same process, interleaved blocks, HIP events, for sparse lane sets and for configs[3]'s own 8 x 24 points (which stays
on the two-launch path)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.draw_heatmap import draw_polylines_multiscale, lanes as lanes_mod  # noqa: E402


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
    B, SH, SW, L, P = 32, 2160, 3840, 8, 24
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)

    def run(maps, strides, fused, **kw):
        lanes_mod.FUSED_SAMPLER = fused
        try:
            draw_polylines_multiscale(maps, run.lanes, 256, 2, strides, clear=True, **kw)
        finally:
            lanes_mod.FUSED_SAMPLER = True

    maps = [torch.empty(B, int(SH / s), int(SW / s), device=dev) for s in (4.0, 8.0, 16.0)]
    for nl, npnt in ((1, 24), (2, 24), (1, 64), (4, 16), (8, 8), (16, 4), (8, 24)):
        idx = torch.linspace(0, P - 1, npnt).round().long()
        xs_, ys_ = xs[:, :L, idx], ys[:, :L, idx]
        if nl > L:     # more, shorter polylines: cut every lane in two
            h = npnt
            big = torch.linspace(0, P - 1, 2 * h).round().long()
            xs_ = torch.cat([xs[:, :, big[:h]], xs[:, :, big[h:]]], 1)
            ys_ = torch.cat([ys[:, :, big[:h]], ys[:, :, big[h:]]], 1)
        run.lanes = torch.stack([xs_[:, :nl], ys_[:, :nl]], -1).contiguous().to(dev)
        strides = (4.0, 8.0, 16.0)
        ref = [torch.empty_like(m) for m in maps]
        run(maps, strides, True)
        k_fused = nat.last_dispatch().split("<")[0]
        run(ref, strides, False)
        same = all(torch.equal(a, b) for a, b in zip(maps, ref))
        best = {"fused": 1e9, "two": 1e9}
        for _ in range(3):
            best["two"] = min(best["two"], gpu_us(lambda: run(maps, strides, False)))
            best["fused"] = min(best["fused"], gpu_us(lambda: run(maps, strides, True)))
        nbytes = sum(m.numel() * 4 for m in maps)
        print(json.dumps({"polylines_per_frame": nl, "points": npnt, "strides": strides, "bit_identical": same, "dispatch": k_fused,
                          "default_path_us": round(best["fused"], 2), "sampler_plus_point_splat_us": round(best["two"], 2),
                          "default_path_frac_of_hbm_peak": round(nbytes / best["fused"] / 1e3 / 8000.0, 3)}), flush=True)


if __name__ == "__main__":
    main()
