"""The per-step target-preparation path end to end, as a training loop would run it on one GPU:

    DataLoader worker      per-frame meta data (boxes, labels, lanes) -> packing_collate: ONE buffer per batch
    multi_tensor_copier    start_copy(packed, "cuda") on the side stream, overlapping the previous step's kernels
    batching_helpers       combine_data: ragged lists -> RaggedBatch (boxes / centres / labels)
    draw_heatmap           draw_targets_multiscale: box maps (strides 4/8/16) + lane maps in two launches
    batching_helpers       batched_bool_indexing / batched_indexing_access on the loss side

Used by tests/test_pipeline_gpu.py, which checks every stage against the CPU oracle; run it directly for timings.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accv-lab_amd"))   # run from a checkout

import time
from typing import Dict, List

import torch

import accvlab.batching_helpers as bh
from accvlab.draw_heatmap import draw_targets_multiscale
from accvlab.multi_tensor_copier import packing_collate, start_copy

STRIDES = (4.0, 8.0, 16.0)
SRC_H, SRC_W = 544, 960
LANES, LANE_POINTS = 4, 12


class Frames(torch.utils.data.Dataset):
    """Synthetic frames: a ragged set of boxes with labels and a fixed number of lanes."""

    def __init__(self, n: int, max_objects: int = 24):
        self.n, self.max_objects = n, max_objects

    def __len__(self):
        return self.n

    def __getitem__(self, i) -> Dict[str, torch.Tensor]:
        g = torch.Generator().manual_seed(1000 + i)
        k = int(torch.randint(0, self.max_objects + 1, (1,), generator=g))
        c = torch.rand(k, 2, generator=g) * torch.tensor([float(SRC_W), float(SRC_H)])
        half = 4.0 + torch.rand(k, 4, generator=g) * 60.0
        boxes = torch.cat([c - half[:, :2], c + half[:, 2:]], 1)
        t = torch.linspace(0, 1, LANE_POINTS).view(1, LANE_POINTS)
        x0 = torch.rand(LANES, 1, generator=g) * SRC_W
        lanes = torch.stack([x0 + (torch.rand(LANES, 1, generator=g) - 0.5) * 300 * t,
                             SRC_H * (1 - 0.95 * t).expand(LANES, LANE_POINTS)], -1)
        return {"idx": i, "centers": c, "boxes": boxes, "labels": torch.randint(0, 5, (k,), generator=g), "lanes": lanes}


def make_loader(n_frames: int, batch: int, workers: int = 2):
    return torch.utils.data.DataLoader(Frames(n_frames), batch_size=batch, num_workers=workers, pin_memory=True,
                                       collate_fn=packing_collate())


def allocate_targets(batch: int, device) -> Dict[str, List[torch.Tensor]]:
    shapes = [(batch, int(SRC_H / s), int(SRC_W / s)) for s in STRIDES]
    return {"objects": [torch.empty(s, device=device) for s in shapes],
            "lanes": [torch.empty(s, device=device) for s in shapes]}


def prepare_targets(samples, targets, device):
    """GPU part of one step; `samples` = list of per-frame dicts already on the GPU.  Returns the ragged ground truth."""
    centers = bh.combine_data([s["centers"] for s in samples])
    boxes = bh.combine_data([s["boxes"] for s in samples], other_with_same_sample_sizes=centers)
    labels = bh.combine_data([s["labels"] for s in samples], other_with_same_sample_sizes=centers)
    lanes = torch.stack([s["lanes"] for s in samples])
    # (= draw_heatmap_multiscale + draw_polylines_multiscale; the polyline sampler rides in the box-map launch)
    draw_targets_multiscale(targets["objects"], centers, boxes, STRIDES, targets["lanes"], lanes, 128, 2, clear=True)
    # loss side: keep the objects of class 0..2 ("vehicles"), compacted, without a host synchronisation
    keep = bh.RaggedBatch(labels.tensor < 3, sample_sizes=labels.sample_sizes)
    vehicles = bh.batched_bool_indexing(boxes, keep, max_sample_size=boxes.tensor.shape[1])
    return centers, boxes, labels, lanes, vehicles


def run(n_frames: int = 64, batch: int = 8, device="cuda:0", workers: int = 2):
    """Yields (samples on the GPU, targets, ragged ground truth) per batch; the copy of batch i+1 is started before the
    kernels of batch i are enqueued, so it overlaps them on the copier's side stream."""
    dev = torch.device(device)
    it = iter(make_loader(n_frames, batch, workers))
    targets = allocate_targets(batch, dev)
    pending = start_copy(next(it), dev)
    while pending is not None:
        samples = pending.get()
        nxt = next(it, None)
        pending = start_copy(nxt, dev) if nxt is not None else None
        yield samples, targets, prepare_targets(samples, targets, dev)


if __name__ == "__main__":
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for samples, targets, _ in run(n_frames=512, batch=32, workers=4):
        n += len(samples)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n} frames in {dt * 1e3:.1f} ms  ({n / dt:.0f} frames/s incl. data loading)")
