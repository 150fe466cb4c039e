#!/usr/bin/env python3
"""The configurations the reference PUBLISHES numbers for, run here on MI355X next to the published figures.
Those were taken on other hardware (A100 / RTX 5000 Ada) — context, not a like-for-like comparison:

  draw_heatmap / draw_heatmap_batched / class-wise   48 maps of 20x50, <= 50 targets per sample   docs/intro.rst:63-85
  multi_tensor_copier                                16 samples, 528 tensors, ~150 KB              docs/evaluation.rst:57-78
  lane_helpers.polyline.interpolate                  batch 64, points x distances grid            evaluation_results/.../batch_64_runtime_cuda.md

Inputs: the draw_heatmap case uses the committed fixture of the reference's own test recipe (tests/golden/h1_g2.npz: seed 7,
48 samples, image 320x800 at stride 16); the other two are synthetic at the published sizes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402


def per_call_us(fn, n=2000, warm=200):
    """back-to-back calls: what a training loop pays per call (the larger of host and device time); measured twice, the faster
    pass reported (the first pass behind a section with many allocations occasionally reads 2x slow)"""
    a = _per_call_us(fn, n, warm)
    b = _per_call_us(fn, n, 0)
    return a if a["wall_us"] <= b["wall_us"] else b


def _per_call_us(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return {"device_us": round(a.elapsed_time(b) / n * 1e3, 2), "wall_us": round((time.perf_counter() - t0) / n * 1e6, 2)}


def main():
    from types import SimpleNamespace

    from accvlab.draw_heatmap import draw_heatmap, draw_heatmap_batched
    from accvlab.lane_helpers.polyline import interpolate
    from accvlab.multi_tensor_copier import start_copy

    dev = torch.device("cuda", 0)
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "h1_g2.npz"))
    h, w, ncls = int(g2["H"]), int(g2["W"]), int(g2["C"])
    centers = torch.from_numpy(g2["centers"]).to(dev)
    radii = torch.from_numpy(g2["radii"]).to(dev)
    labels = torch.from_numpy(g2["labels"]).to(dev)
    sizes = torch.from_numpy(g2["sizes"]).to(dev)
    b = centers.shape[0]
    rb = lambda t: SimpleNamespace(tensor=t, sample_sizes=sizes)  # noqa: E731
    valid = torch.arange(centers.shape[1], device=dev).unsqueeze(0) < sizes.unsqueeze(1)
    flat_c, flat_r = centers[valid].contiguous(), radii[valid].contiguous()
    flat_idx = torch.arange(b, device=dev, dtype=torch.int32).unsqueeze(1).expand_as(radii)[valid].contiguous()
    hm = torch.zeros(b, h, w, device=dev)
    hm_cw = torch.zeros(b, ncls, h, w, device=dev)
    k, factor = float(g2["k"]), float(g2["factor"])
    out = {"draw_heatmap (flat input)": per_call_us(lambda: draw_heatmap(hm, flat_c, flat_r, flat_idx, factor, k)),
           "draw_heatmap_batched": per_call_us(lambda: draw_heatmap_batched(hm, rb(centers), rb(radii), factor, k)),
           "draw_heatmap_batched class-wise": per_call_us(lambda: draw_heatmap_batched(hm_cw, rb(centers), rb(radii), factor, k,
                                                                                      labels=rb(labels)))}
    print(json.dumps({"config": f"draw_heatmap: {b} maps of {h}x{w} ({ncls} classes), <= {centers.shape[1]} targets per sample "
                                "(reference test recipe, seed 7)",
                      "this_build_MI355X": out,
                      "reference_published_A100_ms": {"flat": 0.0482, "batched": 0.0366, "class-wise": 0.059, "pytorch loop": 201.1,
                                                      "source": "packages/draw_heatmap/docs/intro.rst:63-85"}}))

    tree = wl.meta_tensor_tree(528, seed=0)
    sync = torch.cuda.synchronize
    for _ in range(50):
        start_copy(tree, dev).get()
    sync()
    t0 = time.perf_counter()
    for _ in range(500):
        start_copy(tree, dev).get()
    sync()
    t_copier = (time.perf_counter() - t0) / 500
    leaves = []

    def walk(x):
        if isinstance(x, torch.Tensor):
            leaves.append(x)
        elif isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, (list, tuple)):
            for v in x:
                walk(v)

    walk(tree)
    t0 = time.perf_counter()
    for _ in range(50):
        [t.to(dev) for t in leaves]
    sync()
    t_naive = (time.perf_counter() - t0) / 50
    print(json.dumps({"config": f"multi_tensor_copier: {len(leaves)} small CPU tensors, {sum(t.numel() * t.element_size() for t in leaves)} bytes",
                      "this_build_MI355X_ms": {"start_copy().get()": round(t_copier * 1e3, 4), "per-tensor .to()": round(t_naive * 1e3, 4),
                                               "speed-up": round(t_naive / t_copier, 2)},
                      "reference_published_RTX5000Ada_ms": {"start_copy().get()": 0.375, "per-tensor .to()": 3.035, "speed-up": 8.10,
                                                            "source": "packages/multi_tensor_copier/docs/evaluation.rst:57-78"}}))

    published = {1: {(2, 1): 0.003535, (100, 100): 0.003491, (1000, 1000): 0.003689, (2000, 2000): 0.005979, (5000, 5000): 0.0149,
                     (5000, 1): 0.006, (2, 5000): 0.004911, (500, 5000): 0.007429},
                 64: {(2, 1): 0.008093, (100, 100): 0.004731, (1000, 1000): 0.005343, (2000, 2000): 0.007966, (5000, 5000): 0.0223,
                      (5000, 1): 0.007627, (2, 5000): 0.008322, (500, 5000): 0.009575}}
    for batch, cells in published.items():
        rows = {}
        for (pts, dist), ref_ms in cells.items():
            p = torch.rand(batch, pts, 2, device=dev).cumsum(1)
            d = torch.rand(batch, dist, device=dev) * float(pts) * 0.5
            r = per_call_us(lambda: interpolate(p, d), n=1000, warm=100)
            r["published_us"] = round(ref_ms * 1e3, 2)
            rows[f"{pts} points x {dist} distances"] = r
        print(json.dumps({"config": f"lane_helpers.polyline.interpolate, batch {batch}", "this_build_MI355X": rows,
                          "published_on": "RTX 5000 Ada",
                          "source": f"packages/lane_helpers/evaluation_results/polyline_runtime_evaluation/batch_{batch}_runtime_cuda.md"}))

if __name__ == "__main__":
    main()
