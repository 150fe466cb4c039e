"""The whole target-preparation step (examples/target_prep_step.py): DataLoader workers with packing_collate ->
start_copy on the side stream (overlapping the previous batch's kernels) -> combine_data -> multi-scale heat-maps and
lane rasters -> sync-free compaction.  Every stage is checked against the CPU: dataset samples, the oracle maps, torch
indexing — so stream/event ordering mistakes between the copier and the kernels would show up as wrong values."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from oracle import h1 as oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def test_target_prep_step_end_to_end():
    import target_prep_step as step
    from accvlab.draw_heatmap import get_centers_and_radii, sample_lane_targets

    ds = step.Frames(24)
    seen = 0
    for samples, targets, (centers, boxes, labels, lanes, vehicles) in step.run(n_frames=24, batch=8, workers=2):
        b = len(samples)
        idx = [int(s["idx"]) for s in samples]
        assert idx == list(range(seen, seen + b))
        # 1. the copied samples equal the dataset
        for s in samples:
            ref = ds[int(s["idx"])]
            for key in ("centers", "boxes", "labels", "lanes"):
                assert s[key].is_cuda and torch.equal(s[key].cpu(), ref[key])
        # 2. ragged packing
        for i, s in enumerate(samples):
            n = s["boxes"].shape[0]
            assert int(boxes.sample_sizes[i]) == n
            assert torch.equal(boxes.tensor[i, :n], s["boxes"]) and bool((boxes.tensor[i, n:] == 0).all())
        # 3. object maps == oracle on the integer targets of the reference front-end rule
        for hm, stride in zip(targets["objects"], step.STRIDES):
            ci, ri = get_centers_and_radii(centers, boxes, stride)
            want = np.zeros(tuple(hm.shape), dtype=np.float32)
            oracle.draw_heatmap_batched(want, ci.tensor.cpu().numpy(), ri.tensor.cpu().numpy(),
                                        centers.sample_sizes.cpu().numpy(), clear=True)
            assert np.abs(hm.cpu().numpy() - want).max() <= 1e-5
        # 4. lane maps == oracle on the integer targets made from the kernel's own samples
        for hm, stride in zip(targets["lanes"], step.STRIDES):
            ci, ri = sample_lane_targets(lanes, 128, 2, stride)
            want = np.zeros(tuple(hm.shape), dtype=np.float32)
            oracle.draw_heatmap_batched(want, ci.cpu().numpy(), ri.cpu().numpy(),
                                        np.full(b, ci.shape[1], dtype=np.int64), clear=True)
            assert np.abs(hm.cpu().numpy() - want).max() <= 1e-5
            assert float(hm.max()) > 0.5
        # 5. compaction without a host sync == per-sample boolean indexing
        for i, s in enumerate(samples):
            want = s["boxes"][s["labels"] < 3]
            n = int(vehicles.sample_sizes[i])
            assert n == want.shape[0] and torch.equal(vehicles.tensor[i, :n], want)
        seen += b
    assert seen == 24
