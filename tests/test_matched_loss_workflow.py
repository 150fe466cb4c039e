"""SURVEY §8 f3 — the loss-side caller pattern of batching_helpers (cost matrices -> per-frame Hungarian assignment ->
matched gathers -> masked sums -> write-back through the match indices), run end to end through this build's
operators and compared with the plain per-sample loop (examples/matched_loss.py).  The same pattern is what the
reference's packages/batching_helpers/example/{matcher.py,loss_computation.py} exercise.

Bar: identical assignments (integer, exact); losses and gradients equal to 1e-5 relative (fp32 sums in a different
association order)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import matched_loss as ml  # noqa: E402


def _run(device, batch, q, c, max_gt, seed, min_gt=0, fused=False):
    inp = ml.make_inputs(batch, q, c, max_gt, device, seed=seed, min_gt=min_gt)
    gb, gl, gw, pb, ps, pe = inp
    leaves_a = [t.clone().requires_grad_(True) for t in (pb, ps, pe)]
    leaves_b = [t.clone().requires_grad_(True) for t in (pb, ps, pe)]
    la = ml.run_batched(gb, gl, gw, *leaves_a, fused=fused)
    lb = ml.loss_per_sample(gb, gl, gw, *leaves_b)
    assert la.shape == lb.shape == (batch,)
    torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-5)
    la.sum().backward()
    lb.sum().backward()
    for a, b in zip(leaves_a, leaves_b):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-5, atol=1e-6)


def test_indexed_ops_are_gpu_only_like_the_reference():
    # the reference's gather/scatter operators exist only in its CUDA module (batched_indexing_access_cuda.cpp:247-265;
    # the CPU module exports the pad fill alone) -> CPU tensors are refused, never silently computed elsewhere
    inp = ml.make_inputs(2, 6, 3, 4, torch.device("cpu"), seed=0, min_gt=1)
    with pytest.raises(RuntimeError):
        ml.run_batched(*inp)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["operators", "fused-class-and-box-terms"])
@pytest.mark.parametrize("seed", [0, 1])
def test_matched_loss_gpu(seed, fused):
    _run(torch.device("cuda", 0), batch=8, q=100, c=10, max_gt=30, seed=seed, fused=fused)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["operators", "fused-class-and-box-terms"])
def test_matched_loss_gpu_edge_shapes(fused):
    _run(torch.device("cuda", 0), batch=5, q=4, c=3, max_gt=7, seed=11, fused=fused)
    _run(torch.device("cuda", 0), batch=3, q=4, c=3, max_gt=0, seed=12, fused=fused)


def test_match_indices_identical_to_per_frame_assignment():
    import numpy as np
    from scipy.optimize import linear_sum_assignment

    import accvlab.batching_helpers as bh

    gb, gl, gw, pb, ps, pe = ml.make_inputs(7, 16, 4, 10, torch.device("cpu"), seed=5)
    boxes = bh.combine_data(gb)
    labels = bh.combine_data(gl, other_with_same_sample_sizes=boxes)
    m_gt, m_pred = ml.match_batched(boxes, labels, pb, ps)
    assert m_gt.tensor.dtype == torch.int64 and m_pred.tensor.dtype == torch.int64
    for b in range(7):
        cost = (1.0 - ml._iou(pb[b].unsqueeze(1), gb[b].unsqueeze(0))) + (1.0 - ps[b] @ ml._one_hot(gl[b], 4).T)
        rows, cols = linear_sum_assignment(cost.numpy())
        n = int(m_gt.sample_sizes[b])
        assert n == len(rows) == min(16, gb[b].shape[0])
        assert np.array_equal(m_pred.tensor[b, :n].numpy(), rows) and np.array_equal(m_gt.tensor[b, :n].numpy(), cols)
