"""Loss-side consumer of the ragged-batch operators: Hungarian matching + matched losses for a 2-D
detection head, written once in BATCHED form on top of ``accvlab.batching_helpers`` and once as the plain per-sample
loop it replaces.  It is the caller pattern of the reference's packages/batching_helpers/example/
{matcher.py, loss_computation.py} (cost matrices -> per-sample assignment -> matched gather -> masked sums -> existence
loss with weights written back through the match indices); this file is an independent formulation used by
tests/test_matched_loss_workflow.py (batched == per-sample, forward and gradients, CPU and GPU) and timed by
scripts/bench_secondary.py (config F3).

Shapes: ground truth ragged per frame — boxes ``[B, G*, 4]`` (x0,y0,x1,y1), labels ``[B, G*]`` int64, weights
``[B, G*]``; predictions dense — boxes ``[B, Q, 4]``, class scores ``[B, Q, C]``, existence ``[B, Q]``.
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accv-lab_amd"))   # run from a checkout

import torch
from scipy.optimize import linear_sum_assignment

import accvlab.batching_helpers as bh

EPS = 1e-6


# ------------------------------------------------------------------------------------------------ element-wise maths
def _iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """IoU of broadcastable boxes ``[..., 4]`` (union floored at EPS)."""
    wh_a = a[..., 2:] - a[..., :2]
    wh_b = b[..., 2:] - b[..., :2]
    inter_wh = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp(min=0.0)
    inter = inter_wh[..., 0] * inter_wh[..., 1]
    union = (wh_a[..., 0] * wh_a[..., 1] + wh_b[..., 0] * wh_b[..., 1] - inter).clamp(min=EPS)
    return inter / union


def _one_hot(labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    return torch.nn.functional.one_hot(labels.clamp(0, num_classes - 1), num_classes).to(torch.float32)


# ------------------------------------------------------------------------------------------------ batched formulation
def match_batched(gt_boxes: bh.RaggedBatch, gt_labels: bh.RaggedBatch, pred_boxes, pred_scores):
    """-> (matched gt indices, matched prediction indices) as int64 RaggedBatches on the inputs' device."""
    num_classes = pred_scores.shape[-1]
    # padded cost matrices [B, Q, Gmax]: padded gt columns hold junk and are cut off by split() below
    cost = (1.0 - _iou(pred_boxes.unsqueeze(2), gt_boxes.tensor.unsqueeze(1))) + \
           (1.0 - torch.einsum("bqc,bgc->bqg", pred_scores, _one_hot(gt_labels.tensor, num_classes)))
    cost = gt_labels.create_with_sample_sizes_like_self(cost.detach(), non_uniform_dim=2)
    per_frame = cost.to_device(torch.device("cpu")).split()
    gt_idx, pred_idx = [], []
    for m in per_frame:
        rows, cols = linear_sum_assignment(m.numpy())
        pred_idx.append(torch.as_tensor(rows, dtype=torch.int64))
        gt_idx.append(torch.as_tensor(cols, dtype=torch.int64))
    gt_rb = bh.combine_data(gt_idx)
    pred_rb = bh.combine_data(pred_idx, other_with_same_sample_sizes=gt_rb)
    dev = gt_boxes.tensor.device
    return gt_rb.to_device(dev), pred_rb.to_device(dev)


def loss_batched(gt_boxes, gt_labels, gt_weights, pred_boxes, pred_scores, pred_exist, match_gt, match_pred):
    """Per-frame loss ``[B]`` = class L1 + (1 - IoU) over matched pairs + existence L1 over all predictions."""
    num_q, num_classes = pred_scores.shape[1], pred_scores.shape[2]
    lab = bh.batched_indexing_access(gt_labels, match_gt)
    box_g = bh.batched_indexing_access(gt_boxes, match_gt)
    w = bh.batched_indexing_access(gt_weights, match_gt)
    sc = bh.batched_indexing_access(pred_scores, match_pred)
    box_p = bh.batched_indexing_access(pred_boxes, match_pred)

    cls_term = w.tensor * (sc.tensor - _one_hot(lab.tensor.to(torch.int64), num_classes)).abs().sum(-1)
    overlap = _iou(box_g.tensor, box_p.tensor)
    box_term = w.tensor * (1.0 - overlap)
    cls_loss = bh.sum_over_targets(lab.create_with_sample_sizes_like_self(cls_term, non_uniform_dim=1))
    box_loss = bh.sum_over_targets(lab.create_with_sample_sizes_like_self(box_term, non_uniform_dim=1))

    # existence: matched predictions should say 1 (weight = gt weight x overlap), the others 0 with the mean matched
    # weight rescaled by matched / unmatched counts
    matched = bh.get_mask_from_indices(num_q, match_pred)
    w_match = w.create_with_sample_sizes_like_self(w.tensor * (1.0 - box_term), non_uniform_dim=1)
    n = w_match.sample_sizes
    balance = torch.nan_to_num(n / (num_q - n), 0.0)
    background = (bh.average_over_targets(w_match) * balance).unsqueeze(-1).repeat(1, num_q)
    w_all = bh.batched_indexing_write(w_match, match_pred, background)
    exist_loss = (w_all * (pred_exist - matched.to(torch.float32)).abs()).sum(1)
    return cls_loss + box_loss + exist_loss


def loss_batched_fused(gt_boxes, gt_labels, gt_weights, pred_boxes, pred_scores, pred_exist, match_gt, match_pred):
    """The same loss with the class and box terms through ``bh.matched_pair_loss_sum`` (one launch each: the matched
    gathers, the per-object loss — L1 against one-hot labels / 1 - IoU, the two losses of the reference example — the
    weight look-up and the masked per-frame sum).  The existence term keeps the operator composition: it needs the
    PER-OBJECT overlaps, not their sum."""
    num_q = pred_scores.shape[1]
    cls_loss = bh.matched_pair_loss_sum(gt_labels, pred_scores, match_gt, match_pred, gt_weights, kind="onehot_l1")
    box_loss = bh.matched_pair_loss_sum(gt_boxes, pred_boxes, match_gt, match_pred, gt_weights, kind="iou_xyxy", eps=EPS)

    box_g = bh.batched_indexing_access(gt_boxes, match_gt)
    w = bh.batched_indexing_access(gt_weights, match_gt)
    box_p = bh.batched_indexing_access(pred_boxes, match_pred)
    box_term = w.tensor * (1.0 - _iou(box_g.tensor, box_p.tensor))
    matched = bh.get_mask_from_indices(num_q, match_pred)
    w_match = w.create_with_sample_sizes_like_self(w.tensor * (1.0 - box_term), non_uniform_dim=1)
    n = w_match.sample_sizes
    balance = torch.nan_to_num(n / (num_q - n), 0.0)
    background = (bh.average_over_targets(w_match) * balance).unsqueeze(-1).repeat(1, num_q)
    w_all = bh.batched_indexing_write(w_match, match_pred, background)
    exist_loss = (w_all * (pred_exist - matched.to(torch.float32)).abs()).sum(1)
    return cls_loss + box_loss + exist_loss


# ------------------------------------------------------------------------------------------------ per-sample loop
def loss_per_sample(gt_boxes_l, gt_labels_l, gt_weights_l, pred_boxes, pred_scores, pred_exist):
    """The same computation frame by frame with plain tensor indexing (what the batched form replaces)."""
    num_q, num_classes = pred_scores.shape[1], pred_scores.shape[2]
    out = []
    for b, (gb, gl, gw) in enumerate(zip(gt_boxes_l, gt_labels_l, gt_weights_l)):
        pb, ps, pe = pred_boxes[b], pred_scores[b], pred_exist[b]
        cost = (1.0 - _iou(pb.unsqueeze(1), gb.unsqueeze(0))) + (1.0 - ps @ _one_hot(gl, num_classes).T)
        rows, cols = linear_sum_assignment(cost.detach().cpu().numpy())
        rows = torch.as_tensor(rows, dtype=torch.int64, device=pb.device)
        cols = torch.as_tensor(cols, dtype=torch.int64, device=pb.device)
        w = gw[cols]
        cls_term = w * (ps[rows] - _one_hot(gl[cols], num_classes)).abs().sum(-1)
        box_term = w * (1.0 - _iou(gb[cols], pb[rows]))
        w_match = w * (1.0 - box_term)
        n = rows.numel()
        mean_w = w_match.sum() / n if n else w_match.sum() * 0.0
        balance = n / (num_q - n) if num_q != n else 0.0
        w_all = (mean_w * balance).expand(num_q).clone()
        w_all[rows] = w_match
        target = torch.zeros(num_q, device=pb.device)
        target[rows] = 1.0
        out.append(cls_term.sum() + box_term.sum() + (w_all * (pe - target).abs()).sum())
    return torch.stack(out)


# ------------------------------------------------------------------------------------------------ synthetic inputs
def make_inputs(batch, num_queries, num_classes, max_gt, device, seed=0, min_gt=0):
    g = torch.Generator().manual_seed(seed)
    gt_boxes_l, gt_labels_l, gt_weights_l = [], [], []
    for _ in range(batch):
        n = int(torch.randint(min_gt, max_gt + 1, (1,), generator=g))
        tl = torch.rand(n, 2, generator=g) * 80
        gt_boxes_l.append(torch.cat([tl, tl + 4 + torch.rand(n, 2, generator=g) * 30], 1).to(device))
        gt_labels_l.append(torch.randint(0, num_classes, (n,), generator=g).to(device))
        gt_weights_l.append((0.5 + torch.rand(n, generator=g)).to(device))
    tl = torch.rand(batch, num_queries, 2, generator=g) * 80
    pred_boxes = torch.cat([tl, tl + 4 + torch.rand(batch, num_queries, 2, generator=g) * 30], 2).to(device)
    pred_scores = torch.softmax(torch.randn(batch, num_queries, num_classes, generator=g), -1).to(device)
    pred_exist = torch.rand(batch, num_queries, generator=g).to(device)
    return gt_boxes_l, gt_labels_l, gt_weights_l, pred_boxes, pred_scores, pred_exist


def run_batched(gt_boxes_l, gt_labels_l, gt_weights_l, pred_boxes, pred_scores, pred_exist, fused=False):
    gt_boxes = bh.combine_data(gt_boxes_l)
    gt_labels = bh.combine_data(gt_labels_l, other_with_same_sample_sizes=gt_boxes)
    gt_weights = bh.combine_data(gt_weights_l, other_with_same_sample_sizes=gt_boxes)
    m_gt, m_pred = match_batched(gt_boxes, gt_labels, pred_boxes, pred_scores)
    fn = loss_batched_fused if fused else loss_batched          # the fused op is GPU-only
    return fn(gt_boxes, gt_labels, gt_weights, pred_boxes, pred_scores, pred_exist, m_gt, m_pred)
