#!/usr/bin/env python3
"""Generates tests/golden/lane_*.npz by IMPORTING the reference's polyline test oracle (build container only):
packages/lane_helpers/tests/polyline_test_utils.py (sample_batch_cpu, sample_polyline_cpu, polyline_lengths_cpu,
polyline_lengths_var_size_cpu).  That module imports `accvlab.batching_helpers.RaggedBatch`; this repo's drop-in
package provides it.  Inputs follow the reference's tests (test_polyline_fixed_interpolation.py:25-110, 140-190;
test_polyline_var_size_interpolation.py; test_polyline_lengths.py) — literals restated as data, random cases with a
fixed torch.Generator seed."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path[:0] = [os.path.join(ROOT, "accv-lab_amd"), ROOT, "/root/reference/packages/lane_helpers/tests"]

import polyline_test_utils as ref  # noqa: E402  (reference)

out = {}
rect = torch.tensor([[[0.0, 0.0], [1.0, 0.0], [1.0, 2.0], [0.0, 2.0], [0.0, 0.0]]])
d = torch.tensor([[0.0, 0.5, 1.0, 2.0, 3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0, -1.0, 7.5]])
out["rect_points"], out["rect_dist"] = rect.numpy(), d.numpy()
out["rect_expected"] = ref.sample_batch_cpu(rect, d).numpy()
out["rect_length"] = ref.polyline_lengths_cpu(rect).numpy()

offsets = torch.tensor([[0.0, 0.0], [3.5, -1.25]])
pts = rect[0].unsqueeze(0) + offsets.unsqueeze(1)
d2 = torch.tensor([[0.0, 0.5, 1.0, 3.0, 6.0], [6.0, 5.0, 3.0, 1.0, 0.0]])
out["batched_points"], out["batched_dist"] = pts.numpy(), d2.numpy()
out["batched_expected"] = ref.sample_batch_cpu(pts, d2).numpy()

# degenerate: repeated points (zero-length segments), single point
deg = torch.tensor([[[1.0, 1.0, 1.0], [1.0, 1.0, 1.0], [2.0, 1.0, 1.0], [2.0, 1.0, 1.0], [2.0, 3.0, 1.0]]])
dd = torch.tensor([[0.0, 0.25, 1.0, 1.5, 3.0, 9.0]])
out["deg_points"], out["deg_dist"] = deg.numpy(), dd.numpy()
out["deg_expected"] = ref.sample_batch_cpu(deg, dd).numpy()
one = torch.tensor([[[4.0, -2.0]]])
out["one_points"], out["one_dist"] = one.numpy(), torch.tensor([[-1.0, 0.0, 2.0]]).numpy()
out["one_expected"] = ref.sample_batch_cpu(one, torch.tensor([[-1.0, 0.0, 2.0]])).numpy()
out["one_length"] = ref.polyline_lengths_cpu(one).numpy()

g = torch.Generator().manual_seed(0)
for k in range(12):
    npnt = int(torch.randint(15, 61, (), generator=g))
    ndst = int(torch.randint(15, 61, (), generator=g))
    dims = [2, 3, 4][k % 3]
    p = torch.rand((3, npnt, dims), generator=g)
    dist = torch.rand((3, ndst), generator=g)
    tot = torch.linalg.vector_norm(p[:, 1:] - p[:, :-1], dim=2).sum(1)
    dist = dist * tot[:, None] * 1.2 - 0.1 * tot[:, None]       # includes out-of-range queries
    out[f"rand{k}_points"], out[f"rand{k}_dist"] = p.numpy(), dist.numpy()
    out[f"rand{k}_expected"] = ref.sample_batch_cpu(p, dist).numpy()
    out[f"rand{k}_length"] = ref.polyline_lengths_cpu(p).numpy()

# ragged: padded tensors + sizes; expected per sample via the reference per-sample routine
g = torch.Generator().manual_seed(1)
for k in range(6):
    B, P, Q, dims = 7, 12 + 9 * k, 17, [3, 2][k % 2]
    p = torch.rand((B, P, dims), generator=g) * 10
    dist = torch.rand((B, Q), generator=g) * 30 - 2
    ps = torch.randint(0, P + 1, (B,), generator=g)
    qs = torch.randint(0, Q + 1, (B,), generator=g)
    ps[0], qs[1] = 0, 0                                              # empty polyline / no queries
    if k % 2:
        ps[2] = 1
    exp = torch.full((B, Q, dims), float("nan"))
    lens = ref.polyline_lengths_var_size_cpu(p, ps)
    for b in range(B):
        e = ref.sample_polyline_cpu(p[b, : int(ps[b])], dist[b, : int(qs[b])])
        exp[b, : int(qs[b])] = e
    out[f"rag{k}_points"], out[f"rag{k}_dist"] = p.numpy(), dist.numpy()
    out[f"rag{k}_psizes"], out[f"rag{k}_qsizes"] = ps.numpy(), qs.numpy()
    out[f"rag{k}_expected"], out[f"rag{k}_length"] = exp.numpy(), lens.numpy()

np.savez_compressed(os.path.join(HERE, "lane_polyline.npz"), **out)
print("wrote", len(out), "arrays", os.path.getsize(os.path.join(HERE, "lane_polyline.npz")), "bytes")
