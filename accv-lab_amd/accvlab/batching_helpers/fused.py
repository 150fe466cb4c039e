"""(extension) The loss-side caller pattern of this package as ONE launch per direction.

``matched_pair_loss_sum(a, b, matches_a, matches_b, weights, kind)`` equals the composition the reference's example
spells with five gathers, element-wise torch code and two masked sums
(packages/batching_helpers/example/loss_computation.py:37-43, 85-86)::

    ga = batched_indexing_access(a, matches_a)                 # [B, K, *D]
    gb = batched_indexing_access(b, matches_b)
    w  = batched_indexing_access(weights, matches_a)           # [B, K]
    per_object = loss(ga.tensor, gb.tensor).flatten(2).sum(-1) * w.tensor
    out = sum_over_targets(ga.create_with_sample_sizes_like_self(per_object))     # [B]

without materialising the gathered rows; differentiable w.r.t. ``a``, ``b`` and ``weights``.  GPU only, float32.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from .. import _amd_native as _nat
from .ragged import RaggedBatch

_KINDS = {"l1": 0, "l2": 1, "smooth_l1": 2}


def _prep(a, b, ia, ib, counts, weights):
    for name, t in (("data_a", a), ("data_b", b), ("indices_a", ia), ("indices_b", ib), ("nums_indices", counts)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if t.device != a.device:
            raise RuntimeError(f"{name} must be on the same device as data_a")
    if not (a.dtype == torch.float32 and b.dtype == torch.float32):
        raise RuntimeError("matched_pair_loss_sum: float32 data expected")
    if not (ia.dtype in (torch.int32, torch.int64) and ib.dtype == ia.dtype):
        raise RuntimeError("matched_pair_loss_sum: int32 or int64 indices of one dtype expected")
    if counts.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("matched_pair_loss_sum: int32 or int64 sample sizes expected")
    if not (a.dim() >= 2 and b.dim() == a.dim() and ia.dim() == 2 and ib.shape == ia.shape and counts.dim() == 1):
        raise RuntimeError("matched_pair_loss_sum: expected data [B, N, *D], indices [B, K], sample sizes [B]")
    if not (a.size(0) == b.size(0) == ia.size(0) == counts.size(0) and a.shape[2:] == b.shape[2:]):
        raise RuntimeError("matched_pair_loss_sum: batch size / trailing data dimensions differ between the inputs")
    if weights is not None:
        if not (weights.is_cuda and weights.is_contiguous() and weights.dtype == torch.float32
                and weights.shape == a.shape[:2] and weights.device == a.device):
            raise RuntimeError("matched_pair_loss_sum: weights must be float32 [B, N_a] on the same device")


class _MatchedPairLossSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, ia, ib, counts, weights, kind, beta):
        _prep(a, b, ia, ib, counts, weights)
        batch, w_a, w_b, k = a.size(0), a.size(1), b.size(1), ia.size(1)
        row = 1
        for s in a.shape[2:]:
            row *= int(s)
        out = torch.empty((batch,), dtype=torch.float32, device=a.device)
        i64, c64 = int(ia.dtype == torch.int64), int(counts.dtype == torch.int64)
        with _nat.device_guard(a.device):
            _nat.check(_nat.lib().accv_matched_pair_reduce_f32(
                a.data_ptr(), b.data_ptr(), ia.data_ptr(), ib.data_ptr(), counts.data_ptr(),
                weights.data_ptr() if weights is not None else None, batch, w_a, w_b, k, k, row, kind, float(beta), i64, c64,
                out.data_ptr(), _nat.stream_ptr(a.device)), "matched_pair_loss_sum")
        ctx.save_for_backward(a, b, ia, ib, counts, weights if weights is not None else a.new_empty(0))
        ctx.meta = (kind, float(beta), row, weights is not None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        a, b, ia, ib, counts, weights = ctx.saved_tensors
        kind, beta, row, has_w = ctx.meta
        need_a, need_b, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_w and ctx.needs_input_grad[5]
        ga = torch.zeros_like(a) if need_a else None
        gb = torch.zeros_like(b) if need_b else None
        gw = torch.zeros_like(weights) if need_w else None
        if need_a or need_b or need_w:
            grad = grad.contiguous().to(torch.float32)
            i64, c64 = int(ia.dtype == torch.int64), int(counts.dtype == torch.int64)
            with _nat.device_guard(a.device):
                _nat.check(_nat.lib().accv_matched_pair_reduce_bwd_f32(
                    a.data_ptr(), b.data_ptr(), ia.data_ptr(), ib.data_ptr(), counts.data_ptr(),
                    weights.data_ptr() if has_w else None, grad.data_ptr(), a.size(0), a.size(1), b.size(1), ia.size(1),
                    ia.size(1), row, kind, beta, i64, c64, ga.data_ptr() if need_a else None,
                    gb.data_ptr() if need_b else None, gw.data_ptr() if need_w else None, _nat.stream_ptr(a.device)),
                    "matched_pair_loss_sum backward")
        return ga, gb, None, None, None, gw, None, None


def matched_pair_loss_sum(data_a, data_b, indices_a: RaggedBatch, indices_b: RaggedBatch,
                          weights: Optional[torch.Tensor] = None, kind: str = "l1", beta: float = 1.0) -> torch.Tensor:
    """Per-sample sum over the matched pairs of an element-wise loss between ``data_a[i, indices_a[i, j]]`` and
    ``data_b[i, indices_b[i, j]]`` (``j < indices_a.sample_sizes[i]``), optionally weighted by
    ``weights[i, indices_a[i, j]]``.

    Args:
        data_a, data_b: float32 ``[B, N_a, *D]`` / ``[B, N_b, *D]`` tensors or RaggedBatch instances (their padding is
            never read: only matched rows are).
        indices_a, indices_b: RaggedBatch int32/int64 ``[B, K]`` with identical sample sizes (the two sides of a matching).
        weights: optional float32 ``[B, N_a]`` (tensor or RaggedBatch), looked up through ``indices_a``.
        kind: ``"l1"`` (|d|), ``"l2"`` (d^2) or ``"smooth_l1"`` (``torch.nn.functional.smooth_l1_loss`` with ``beta``),
            summed over the trailing data dimensions.

    Returns: float32 ``[B]``.  One kernel launch forward, one backward.
    """
    if kind not in _KINDS:
        raise RuntimeError(f"kind must be one of {sorted(_KINDS)}")
    a = data_a.tensor if isinstance(data_a, RaggedBatch) else data_a
    b = data_b.tensor if isinstance(data_b, RaggedBatch) else data_b
    w = weights.tensor if isinstance(weights, RaggedBatch) else weights
    assert indices_a.num_batch_dims == 1 and indices_b.num_batch_dims == 1, "one batch dimension expected"
    assert indices_a.tensor.shape == indices_b.tensor.shape, "the two index batches must have the same shape"
    return _MatchedPairLossSum.apply(a.contiguous(), b.contiguous(), indices_a.tensor.contiguous(), indices_b.tensor.contiguous(),
                                     indices_a.sample_sizes.contiguous(), w.contiguous() if w is not None else None,
                                     _KINDS[kind], beta)
