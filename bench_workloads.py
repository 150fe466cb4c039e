"""Synthetic workloads named in BASELINE.json `configs` / SURVEY.md §8(d). Shared by bench.py and tests.

Everything is generated on the CPU with an explicit torch.Generator so the GPU box and the build
container see identical inputs. No reference code is imported here.
"""
from __future__ import annotations

import torch


def heatmap_objects(batch: int, height: int, width: int, n_min: int = 1, n_max: int = 128, rule: str = "A",
                    seed: int = 42, n_classes: int = 0):
    """C1-style ragged objects (radius rule of the reference benchmark,
    packages/draw_heatmap/benchmark/benchmark.py:56-70).

    Returns python lists (one entry per sample) of int32 tensors: centers (n,2) as (x, y), radii (n,)
    and, when ``n_classes`` > 0, labels (n,).
    rule "A": r = int(clamp(U * min(H,W)/4, min=1))   (r in [1, min(H,W)/4])
    rule "B": r = randint(1, 33)                         (detection-head-like small objects)
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    centers, radii, labels = [], [], []
    for _ in range(batch):
        n = int(torch.randint(n_min, n_max + 1, (1,), generator=g))
        cx = (torch.rand(n, generator=g) * width).to(torch.int32)
        cy = (torch.rand(n, generator=g) * height).to(torch.int32)
        if rule == "A":
            r = torch.clamp(torch.rand(n, generator=g) * (min(height, width) / 4.0), min=1.0).to(torch.int32)
        elif rule == "B":
            r = torch.randint(1, 33, (n,), generator=g, dtype=torch.int32)
        else:
            raise ValueError(f"unknown radius rule {rule!r}")
        centers.append(torch.stack([cx, cy], dim=1).contiguous())
        radii.append(r)
        if n_classes > 0:
            labels.append(torch.randint(0, n_classes, (n,), generator=g, dtype=torch.int32))
    if n_classes > 0:
        return centers, radii, labels
    return centers, radii


def pad_ragged(items, pad_value=0):
    """Minimal list -> (padded tensor, int64 sizes) packer used where batching_helpers must not be a
    dependency of the thing under test."""
    b = len(items)
    n_max = max([int(t.shape[0]) for t in items] + [1])
    inner = tuple(items[0].shape[1:])
    out = torch.full((b, n_max) + inner, pad_value, dtype=items[0].dtype)
    sizes = torch.zeros(b, dtype=torch.int64)
    for i, t in enumerate(items):
        out[i, : t.shape[0]] = t
        sizes[i] = t.shape[0]
    return out, sizes


def ragged_boxes(batch: int = 64, n_min: int = 1, n_max: int = 32, seed: int = 0):
    """C0: `batch` CPU tensors of shape (n_i, 4) fp32."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return [torch.rand(int(torch.randint(n_min, n_max + 1, (1,), generator=g)), 4, generator=g)
            for _ in range(batch)]


def meta_tensor_tree(num_tensors: int = 10_000, seed: int = 0):
    """C2: nested list-of-dicts-of-lists of small CPU tensors, fp32/int64 alternating, shapes drawn from
    {(n,4),(n,),(3,4),(n,7)}, n in [1,128] (the family used by the reference example,
    packages/multi_tensor_copier/example/example.py:25-58)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    leaves = []
    for i in range(num_tensors):
        n = int(torch.randint(1, 129, (1,), generator=g))
        kind = int(torch.randint(0, 4, (1,), generator=g))
        shape = [(n, 4), (n,), (3, 4), (n, 7)][kind]
        if i % 2 == 0:
            leaves.append(torch.rand(shape, generator=g))
        else:
            leaves.append(torch.randint(0, 1000, shape, generator=g, dtype=torch.int64))
    per_sample = 20
    tree = []
    for s in range(0, num_tensors, per_sample):
        chunk = leaves[s:s + per_sample]
        half = len(chunk) // 2
        tree.append({"gt": chunk[:half], "meta": {"aux": tuple(chunk[half:]), "id": s // per_sample,
                                                  "name": f"sample_{s // per_sample}"}})
    return tree
