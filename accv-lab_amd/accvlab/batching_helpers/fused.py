"""(extension) The loss-side caller pattern of this package as ONE launch per direction.

``matched_pair_loss_sum(a, b, matches_a, matches_b, weights, kind)`` equals the composition the reference's example
spells with five gathers, element-wise torch code and two masked sums
(packages/batching_helpers/example/loss_computation.py:37-43, 85-86)::

    ga = batched_indexing_access(a, matches_a)                 # [B, K, *D]
    gb = batched_indexing_access(b, matches_b)
    w  = batched_indexing_access(weights, matches_a)           # [B, K]
    per_object = loss(ga.tensor, gb.tensor).flatten(2).sum(-1) * w.tensor
    out = sum_over_targets(ga.create_with_sample_sizes_like_self(per_object))     # [B]

without materialising the gathered rows; differentiable w.r.t. ``a``, ``b`` and ``weights``.  GPU only; float32, float16,
bfloat16 and float64 data; besides the element-wise kinds the two per-object losses of that example (IoU overlap of boxes,
L1 between one-hot labels and scores).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd.function import once_differentiable

from .. import _amd_native as _nat
from .ragged import RaggedBatch

_KINDS = {"l1": 0, "l2": 1, "smooth_l1": 2, "iou_xyxy": 3, "onehot_l1": 4}
_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}
_ONEHOT = _KINDS["onehot_l1"]


def _prep(a, b, ia, ib, counts, weights, kind):
    for name, t in (("data_a", a), ("data_b", b), ("indices_a", ia), ("indices_b", ib), ("nums_indices", counts)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda):
            raise RuntimeError(f"{name} must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if t.device != a.device:
            raise RuntimeError(f"{name} must be on the same device as data_a")
    if b.dtype not in _DTYPES:
        raise RuntimeError("matched_pair_loss_sum: float32, float16, bfloat16 or float64 data expected")
    if kind == _ONEHOT:
        if a.dtype not in (torch.int32, torch.int64):
            raise RuntimeError("matched_pair_loss_sum: kind 'onehot_l1' takes int32 or int64 class labels as data_a")
        if not (a.dim() == 2 and b.dim() == 3):
            raise RuntimeError("matched_pair_loss_sum: kind 'onehot_l1' expects labels [B, N_a] and scores [B, N_b, C]")
    elif a.dtype != b.dtype:
        raise RuntimeError("matched_pair_loss_sum: data_a and data_b must have the same dtype")
    if not (ia.dtype in (torch.int32, torch.int64) and ib.dtype == ia.dtype):
        raise RuntimeError("matched_pair_loss_sum: int32 or int64 indices of one dtype expected")
    if counts.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("matched_pair_loss_sum: int32 or int64 sample sizes expected")
    if not (a.dim() >= 2 and (kind == _ONEHOT or b.dim() == a.dim()) and ia.dim() == 2 and ib.shape == ia.shape and counts.dim() == 1):
        raise RuntimeError("matched_pair_loss_sum: expected data [B, N, *D], indices [B, K], sample sizes [B]")
    if not (a.size(0) == b.size(0) == ia.size(0) == counts.size(0) and (kind == _ONEHOT or a.shape[2:] == b.shape[2:])):
        raise RuntimeError("matched_pair_loss_sum: batch size / trailing data dimensions differ between the inputs")
    if kind == _KINDS["iou_xyxy"] and tuple(a.shape[2:]) != (4,):
        raise RuntimeError("matched_pair_loss_sum: kind 'iou_xyxy' expects boxes [B, N, 4] as (x0, y0, x1, y1)")
    if weights is not None:
        if not (weights.is_cuda and weights.is_contiguous() and weights.dtype == b.dtype
                and weights.shape == a.shape[:2] and weights.device == a.device):
            raise RuntimeError("matched_pair_loss_sum: weights must be [B, N_a] of the data dtype on the same device")


def _flags(a, ia, counts, kind):
    return ((_nat.MP_IDX_I64 if ia.dtype == torch.int64 else 0) | (_nat.MP_COUNTS_I64 if counts.dtype == torch.int64 else 0)
            | (_nat.MP_LABELS_I64 if kind == _ONEHOT and a.dtype == torch.int64 else 0))


class _MatchedPairLossSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, ia, ib, counts, weights, kind, beta, eps):
        _prep(a, b, ia, ib, counts, weights, kind)
        batch, w_a, w_b, k = a.size(0), a.size(1), b.size(1), ia.size(1)
        row = 1
        for s in b.shape[2:]:
            row *= int(s)
        acc_dtype = torch.float64 if b.dtype == torch.float64 else torch.float32
        out = torch.empty((batch,), dtype=acc_dtype, device=a.device)
        flags = _flags(a, ia, counts, kind)
        with _nat.device_guard(a.device):
            _nat.check(_nat.lib().accv_matched_pair_reduce(
                a.data_ptr(), b.data_ptr(), ia.data_ptr(), ib.data_ptr(), counts.data_ptr(),
                weights.data_ptr() if weights is not None else None, batch, w_a, w_b, k, k, row, kind, _DTYPES[b.dtype],
                float(beta), float(eps), flags, out.data_ptr(), _nat.stream_ptr(a.device)), "matched_pair_loss_sum")
        ctx.save_for_backward(a, b, ia, ib, counts, weights if weights is not None else b.new_empty(0))
        ctx.meta = (kind, float(beta), float(eps), row, weights is not None, flags)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        a, b, ia, ib, counts, weights = ctx.saved_tensors
        kind, beta, eps, row, has_w, flags = ctx.meta
        need_a = ctx.needs_input_grad[0] and kind != _ONEHOT
        need_b, need_w = ctx.needs_input_grad[1], has_w and ctx.needs_input_grad[5]
        # gradients accumulate in the arithmetic type (float atomics; half-precision atomics would round per add) and are
        # cast to the data dtype once
        acc_dtype = torch.float64 if b.dtype == torch.float64 else torch.float32
        ga = torch.zeros(a.shape, dtype=acc_dtype, device=a.device) if need_a else None
        gb = torch.zeros(b.shape, dtype=acc_dtype, device=b.device) if need_b else None
        gw = torch.zeros(weights.shape, dtype=acc_dtype, device=b.device) if need_w else None
        if need_a or need_b or need_w:
            grad = grad.contiguous().to(acc_dtype)
            with _nat.device_guard(a.device):
                _nat.check(_nat.lib().accv_matched_pair_reduce_bwd(
                    a.data_ptr(), b.data_ptr(), ia.data_ptr(), ib.data_ptr(), counts.data_ptr(),
                    weights.data_ptr() if has_w else None, grad.data_ptr(), a.size(0), a.size(1), b.size(1), ia.size(1),
                    ia.size(1), row, kind, _DTYPES[b.dtype], beta, eps, flags, ga.data_ptr() if need_a else None,
                    gb.data_ptr() if need_b else None, gw.data_ptr() if need_w else None, _nat.stream_ptr(a.device)),
                    "matched_pair_loss_sum backward")
        cast = (lambda t: t if t is None or t.dtype == b.dtype else t.to(b.dtype))
        return cast(ga), cast(gb), None, None, None, cast(gw), None, None, None


def matched_pair_loss_sum(data_a, data_b, indices_a: RaggedBatch, indices_b: RaggedBatch,
                          weights: Optional[torch.Tensor] = None, kind: str = "l1", beta: float = 1.0,
                          eps: float = 1e-6) -> torch.Tensor:
    """Per-sample sum over the matched pairs of a per-object loss between ``data_a[i, indices_a[i, j]]`` and
    ``data_b[i, indices_b[i, j]]`` (``j < indices_a.sample_sizes[i]``), optionally weighted by
    ``weights[i, indices_a[i, j]]``.

    Args:
        data_a, data_b: ``[B, N_a, *D]`` / ``[B, N_b, *D]`` tensors or RaggedBatch instances of one dtype — float32,
            float16, bfloat16 or float64, the dtypes the gathers accept (their padding is never read: only matched rows are).
        indices_a, indices_b: RaggedBatch int32/int64 ``[B, K]`` — the two sides of a matching.  ONLY
            ``indices_a.sample_sizes`` is read: slot ``j`` of sample ``i`` is a pair iff ``j < indices_a.sample_sizes[i]``,
            and ``indices_b`` must hold a valid index in each of those slots.  The sample sizes of ``indices_b`` are not
            consulted (comparing the two would cost a device synchronisation per call); a matcher produces both sides
            with the same sizes.
        weights: optional ``[B, N_a]`` of the data dtype (tensor or RaggedBatch), looked up through ``indices_a``.
        kind: element-wise, summed over the trailing data dimensions — ``"l1"`` (|d|), ``"l2"`` (d^2), ``"smooth_l1"``
            (``torch.nn.functional.smooth_l1_loss`` with ``beta``); or one of the per-object losses of the reference's
            example (``packages/batching_helpers/example/loss_computation.py``) — ``"iou_xyxy"``: 1 - IoU of
            ``(x0, y0, x1, y1)`` boxes, union clamped to ``eps`` (``_per_object_bbox_overlap_loss``, :240-274);
            ``"onehot_l1"``: ``data_a`` holds integer class labels ``[B, N_a]``, ``data_b`` scores ``[B, N_b, C]``,
            loss = sum_c |onehot(label)[c] - score[c]| (``_per_object_class_l1_loss_labels_gt``, :225-238).

    Returns: ``[B]`` in the arithmetic dtype — float32, or float64 for float64 data (half-precision data are converted on
    load and accumulated in float32).  One kernel launch forward, one backward (+ one cast for half-precision gradients).
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
                                     _KINDS[kind], beta, eps)
