#!/usr/bin/env python3
"""Generates tests/golden/h1_*.npz by IMPORTING the reference's python oracle (build container only).

Runs only where /root/reference exists; the committed .npz files are what travels to the GPU box.
Imported from the reference (never copied): packages/draw_heatmap/tests/_gaussian_drawing.py
(draw_heatmap_gaussian) and packages/draw_heatmap/tests/_test_helpers.py (generate_gt_bboxes,
generate_gt_bboxes_with_labels, get_centers_and_radii, get_heatmap_single[_with_labels]).

  G1  literals of packages/draw_heatmap/examples/input_data.py:23-59 (restated as data; that module
      itself hard-codes device="cuda:0" and cannot be imported here)
  G2  recipe of tests/test_draw_heatmap*.py:19-55: torch.manual_seed(7), 48 maps of 20x50, <=50 targets,
      stride 16, k=0.8 (+ 20-class variant); padded slots hold drawable junk
      (test_draw_heatmap_batched.py:50-58)
  G3  edge cases written by this build (borders, outside the frame, huge radius, coincident objects,
      empty samples, k != 1, factor in {3, 6, 12}, negative base values)
  G4  C1-shaped: 2 frames 1080x1920, N in [1,128], radius rule A, + 1 frame rule B; stores inputs,
      per-64x64-tile CRC32 + float64 tile sums of the reference map, and 16 full tiles per frame
"""
import os
import sys
import zlib

import numpy as np
import torch

REF = "/root/reference/packages/draw_heatmap/tests"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from _gaussian_drawing import draw_heatmap_gaussian  # noqa: E402  (reference)
import _test_helpers as rh  # noqa: E402  (reference)

import bench_workloads as wl  # noqa: E402


def ref_draw(hm, centers, radii, k, factor):
    """hm: torch [H,W] fp32 (modified in place) via the reference per-object routine."""
    for c, r in zip(centers.tolist(), radii.tolist()):
        draw_heatmap_gaussian(hm, c, int(r), k, factor)
    return hm


def pad(items, fill):
    t, sizes = wl.pad_ragged(items, fill)
    return t.numpy(), sizes.numpy()


def g1():
    centers = [[[2, 3], [67, 50], [21, 10]], [[4, 65], [54, 1]], [[5, 76]], [[76, 13]]]
    radii = [[4, 1, 5], [1, 10], [5], [9]]
    labels = [[4, 10, 2], [3, 1], [0], [7]]
    H = W = 100
    C = 20
    cl = [torch.tensor(c, dtype=torch.int32) for c in centers]
    rl = [torch.tensor(r, dtype=torch.int32) for r in radii]
    ll = [torch.tensor(l, dtype=torch.int32) for l in labels]
    single = torch.stack(rh.get_heatmap_multiple_samples(cl, [r.tolist() for r in rl], 1, 6, [H, W]))
    cw = torch.stack(rh.get_heatmaps_multiple_samples_with_labels(cl, [r.tolist() for r in rl], ll, 1, 6, [H, W], C))
    nz = [(s, c) for s in range(4) for c in range(C) if cw[s, c].abs().sum() > 0]
    c_pad, sizes = pad(cl, 0)
    r_pad, _ = pad(rl, 0)
    l_pad, _ = pad(ll, 0)
    np.savez_compressed(os.path.join(HERE, "h1_g1.npz"), centers=c_pad, radii=r_pad, labels=l_pad, sizes=sizes,
                        H=H, W=W, C=C, k=1.0, factor=6.0, expected=single.numpy(),
                        cw_planes=np.array(nz, dtype=np.int32),
                        cw_expected=np.stack([cw[s, c].numpy() for s, c in nz]))
    # known-answer observed in SURVEY §8c
    assert single[0, 3, 2] == 1.0 and abs(float(single[0, 0, 0]) - 0.05563799664378166) < 1e-9


def g2():
    B, NMAX, IMG, BOX, STRIDE, HM, C = 48, 50, [320, 800, 3], 120, 16, [20, 50], 20
    k, factor = 0.8, 6
    torch.manual_seed(7)
    cen_l, box_l, lab_l = rh.generate_gt_bboxes_with_labels("cpu", B, NMAX, IMG, BOX, C)
    # padded tensors with drawable junk in the padding (test_draw_heatmap_batched.py:50-58)
    cen = torch.ones((B, NMAX, 2)) * 2
    box = torch.zeros((B, NMAX, 4))
    box[:, :, 0:2] = 1
    box[:, :, 2:4] = 3
    lab = torch.zeros((B, NMAX), dtype=torch.int32)
    sizes = torch.zeros(B, dtype=torch.int64)
    for i in range(B):
        n = cen_l[i].shape[0]
        cen[i, :n] = cen_l[i]
        box[i, :n] = box_l[i]
        lab[i, :n] = lab_l[i].to(torch.int32)
        sizes[i] = n
    ci_l, ri_l = rh.get_centers_and_radii_multiple_samples(cen_l, box_l, STRIDE)
    ri_py = [t.numpy().tolist() for t in ri_l]
    single = torch.stack(rh.get_heatmap_multiple_samples(ci_l, ri_py, k, factor, HM))
    cw = torch.stack(rh.get_heatmaps_multiple_samples_with_labels(ci_l, ri_py, lab_l, k, factor, HM, C))
    ci, ri = rh.get_centers_and_radii(cen, box, STRIDE)
    np.savez_compressed(os.path.join(HERE, "h1_g2.npz"), centers=ci.numpy(), radii=ri.numpy(), labels=lab.numpy(),
                        sizes=sizes.numpy(), H=HM[0], W=HM[1], C=C, k=k, factor=float(factor),
                        expected=single.numpy(), cw_expected=cw.numpy(),
                        # target-prep front end (f2): float inputs of get_centers_and_radii
                        centers_f=cen.numpy(), boxes_f=box.numpy(), stride=STRIDE)


def g3():
    cases = []

    def case(name, H, W, objs, k=1.0, factor=6.0, base=0.0):
        hm = torch.full((H, W), float(base))
        c = torch.tensor([[o[0], o[1]] for o in objs], dtype=torch.int32).reshape(-1, 2)
        r = torch.tensor([o[2] for o in objs], dtype=torch.int32)
        ref_draw(hm, c, r, k, factor)
        cases.append((name, H, W, c.numpy(), r.numpy(), k, factor, base, hm.numpy()))

    case("corners", 32, 48, [(0, 0, 3), (47, 0, 5), (0, 31, 4), (47, 31, 6)])
    case("borders", 40, 64, [(0, 20, 7), (63, 20, 7), (30, 0, 9), (30, 39, 9)])
    case("outside", 24, 24, [(-3, 5, 6), (26, 5, 6), (5, -2, 4), (5, 27, 5), (-40, -40, 3), (100, 3, 2)])
    case("r1", 16, 16, [(8, 8, 1), (0, 0, 1), (15, 15, 1)])
    case("huge_radius", 20, 28, [(10, 9, 40)])
    case("coincident", 32, 32, [(16, 16, 5), (16, 16, 5), (16, 16, 9), (17, 16, 2)])
    case("k_small", 32, 32, [(10, 10, 6), (20, 18, 8)], k=0.37)
    case("k_large", 32, 32, [(10, 10, 6), (20, 18, 8)], k=2.5)
    case("factor3", 48, 48, [(20, 22, 11), (30, 8, 6)], factor=3.0)
    case("factor12", 48, 48, [(20, 22, 11), (30, 8, 6)], factor=12.0)
    case("neg_base", 24, 40, [(12, 12, 5), (30, 6, 9)], base=-1.0)
    case("k_negative", 24, 40, [(12, 12, 5)], k=-0.5, base=-1.0)
    case("base_high", 24, 40, [(12, 12, 5), (30, 6, 9)], base=0.5)
    case("wide_unaligned", 37, 131, [(5, 5, 9), (64, 18, 20), (130, 36, 12), (128, 0, 3)])
    case("aligned_tiles", 48, 256, [(0, 0, 2), (127, 15, 6), (128, 16, 6), (255, 47, 30), (100, 24, 70)])
    out = {}
    for name, H, W, c, r, k, factor, base, hm in cases:
        out[f"{name}__centers"] = c
        out[f"{name}__radii"] = r
        out[f"{name}__meta"] = np.array([H, W, k, factor, base], dtype=np.float64)
        out[f"{name}__expected"] = hm
    np.savez_compressed(os.path.join(HERE, "h1_g3.npz"), **out)


def g4():
    H, W, T = 1080, 1920, 64
    out = {"H": H, "W": W, "tile": T}
    frames = []
    cA, rA = wl.heatmap_objects(2, H, W, 1, 128, "A", seed=42)
    cB, rB = wl.heatmap_objects(1, H, W, 1, 128, "B", seed=43)
    frames = [("A0", cA[0], rA[0]), ("A1", cA[1], rA[1]), ("B0", cB[0], rB[0])]
    rng = np.random.RandomState(0)
    for name, c, r in frames:
        hm = torch.zeros(H, W)
        ref_draw(hm, c, r, 1.0, 6.0)
        a = hm.numpy()
        ty, tx = (H + T - 1) // T, W // T
        crc = np.zeros((ty, tx), dtype=np.uint32)
        sums = np.zeros((ty, tx), dtype=np.float64)
        for i in range(ty):
            for j in range(tx):
                t = np.ascontiguousarray(a[i * T:(i + 1) * T, j * T:(j + 1) * T])
                crc[i, j] = zlib.crc32(t.tobytes())
                sums[i, j] = t.astype(np.float64).sum()
        picks = rng.choice(ty * tx, 16, replace=False)
        # bias the picks toward busy tiles: take the 8 densest + 8 random
        dense = np.argsort(-sums.reshape(-1))[:8]
        picks = np.unique(np.concatenate([dense, picks[:8]]))
        tiles = np.stack([a[(p // tx) * T:(p // tx + 1) * T, (p % tx) * T:(p % tx + 1) * T] for p in picks
                          if (p // tx + 1) * T <= H])
        picks = np.array([p for p in picks if (p // tx + 1) * T <= H], dtype=np.int32)
        out[f"{name}__centers"] = c.numpy()
        out[f"{name}__radii"] = r.numpy()
        out[f"{name}__crc"] = crc
        out[f"{name}__sums"] = sums
        out[f"{name}__picks"] = picks
        out[f"{name}__tiles"] = tiles
        print(name, "objects", len(r), "sum", float(a.sum()), "nonzero frac", float((a > 0).mean()))
    np.savez_compressed(os.path.join(HERE, "h1_g4.npz"), **out)


if __name__ == "__main__":
    g1()
    g2()
    g3()
    g4()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
