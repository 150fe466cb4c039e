#!/usr/bin/env python3
"""Secondary configs of BASELINE.json (not the headline metric; bench.py owns that):

  C0  batching_helpers pack -> mask -> split on torch-CPU, 64 samples, N in [1,32], (n,4) fp32   [us / op]
  C2  multi_tensor_copier: 10k mixed fp32/int64 small CPU tensors, nested -> GPU, vs per-tensor .to() and a generic
      recursive .to()                                                                           [ms / batch, GB/s]
  C3  multi-scale heat-maps (strides 4/8/16 of a 3840x2160 source), batch 32, bbox -> centre/radius front end
      + fused clear+draw per scale                                                             [frames/s]

Prints one JSON object per config.  C2/C3 need a GPU; C0 runs anywhere.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402


def _timeit(fn, warm, iters, sync=None):
    for _ in range(warm):
        fn()
    if sync:
        sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    if sync:
        sync()
    return (time.perf_counter() - t0) / iters


def _graph_time(fn, warm=20, iters=500):
    """ms per replay of `fn` captured into one hipGraph (GPU-side cost without the python/launch overhead)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return _timeit(g.replay, warm, iters, torch.cuda.synchronize)


def c0():
    from accvlab.batching_helpers import RaggedBatch, combine_data

    boxes = wl.ragged_boxes(64, 1, 32, seed=0)

    def ours():
        rb = combine_data(boxes)
        _ = rb.mask
        return rb.split()

    def loop_reference_style():
        # the reference's python structure (one slice-assign per sample; per-sample size read in split),
        # restated for timing only: batched_processing_py.py:410-427, ragged_batch.py:870-934
        n = max(b.shape[0] for b in boxes)
        data = torch.zeros((len(boxes), n, 4))
        sizes = torch.empty(len(boxes), dtype=torch.int64)
        for i, b in enumerate(boxes):
            sizes[i] = b.shape[0]
            data[i, : b.shape[0]] = b
        mask = torch.arange(n).unsqueeze(0) < sizes.unsqueeze(1)
        return [data[i][: sizes[i]] for i in range(len(boxes))], mask

    t_ours = _timeit(ours, 20, 200)
    t_loop = _timeit(loop_reference_style, 20, 200)
    print(json.dumps({"config": "C0", "metric": "pack+mask+split, 64 CPU samples (n,4) fp32", "ours_us": t_ours * 1e6,
                      "per_sample_loop_us": t_loop * 1e6, "speedup": t_loop / t_ours}))


def _to_recursive(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev, non_blocking=True)
    if isinstance(x, dict):
        return {k: _to_recursive(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_recursive(v, dev) for v in x)
    return x


def _leaves(x):
    if isinstance(x, torch.Tensor):
        return [x]
    if isinstance(x, dict):
        return [l for v in x.values() for l in _leaves(v)]
    if isinstance(x, (list, tuple)):
        return [l for v in x for l in _leaves(v)]
    return []


def c2(num_tensors):
    from accvlab.multi_tensor_copier import start_copy

    dev = torch.device("cuda", 0)
    tree = wl.meta_tensor_tree(num_tensors, seed=0)
    leaves = _leaves(tree)
    nbytes = sum(t.numel() * t.element_size() for t in leaves)
    sync = torch.cuda.synchronize

    def naive():
        return [t.to(dev) for t in leaves]

    t_naive = _timeit(naive, 3, 10, sync)
    t_rec = _timeit(lambda: _to_recursive(tree, dev), 3, 10, sync)
    t_mtc = _timeit(lambda: start_copy(tree, dev).get(), 10, 50, sync)
    t_mtc_inline = _timeit(lambda: start_copy(tree, dev, use_background_thread=False).get(), 5, 30, sync)
    # latency hidden behind other work: time only start_copy() itself
    handles = []
    t_submit = _timeit(lambda: handles.append(start_copy(tree, dev)), 2, 20)
    for h in handles:
        h.get()
    print(json.dumps({"config": "C2", "tensors": len(leaves), "bytes": nbytes, "per_tensor_to_ms": t_naive * 1e3,
                      "recursive_to_ms": t_rec * 1e3, "multi_tensor_copier_ms": t_mtc * 1e3,
                      "multi_tensor_copier_inline_ms": t_mtc_inline * 1e3, "start_copy_call_ms": t_submit * 1e3,
                      "speedup_vs_per_tensor": t_naive / t_mtc, "speedup_vs_recursive": t_rec / t_mtc,
                      "effective_GBps": nbytes / t_mtc / 1e9}))


def c2_reverse(num_tensors):
    """the same tree, but resident on the GPU and copied back to the host (the reference copies every
    tensor on its own, multi_tensor_copier.cpp:790-800; here: one gather kernel + one transfer per chunk)."""
    from accvlab.multi_tensor_copier import start_copy

    dev = torch.device("cuda", 0)
    tree = start_copy(wl.meta_tensor_tree(num_tensors, seed=0), dev).get()
    leaves = _leaves(tree)
    nbytes = sum(t.numel() * t.element_size() for t in leaves)
    sync = torch.cuda.synchronize
    t_naive = _timeit(lambda: [t.cpu() for t in leaves], 2, 5, sync)
    t_mtc = _timeit(lambda: start_copy(tree, "cpu").get(), 5, 30, sync)
    t_nopack = _timeit(lambda: start_copy(tree, "cpu", pack_cpu_tensors=False).get(), 2, 5, sync)
    print(json.dumps({"config": "C2-reverse (GPU->host)", "tensors": len(leaves), "bytes": nbytes,
                      "per_tensor_cpu_ms": t_naive * 1e3, "multi_tensor_copier_ms": t_mtc * 1e3,
                      "multi_tensor_copier_unpacked_ms": t_nopack * 1e3, "speedup_vs_per_tensor": t_naive / t_mtc}))


def c3():
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_batched

    dev = torch.device("cuda", 0)
    B, SH, SW = 32, 2160, 3840
    g = torch.Generator().manual_seed(7)
    centers_f, boxes_f = [], []
    for _ in range(B):
        n = int(torch.randint(1, 129, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([SW, SH])
        half = torch.rand(n, 4, generator=g) * 400
        centers_f.append(c)
        boxes_f.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
    scales = []
    for s in (4, 8, 16):
        # front end of packages/draw_heatmap/tests/_test_helpers.py:20-28: r = max(1, ceil(min edge dist / s)), c = int(c / s)
        cl, rl = [], []
        for c, b in zip(centers_f, boxes_f):
            d = torch.cat([c - b[:, :2], b[:, 2:] - c], 1)
            r = torch.ceil(d.min(1)[0] / s).to(torch.int32).clamp(min=1)
            cl.append((c / s).to(torch.int32))
            rl.append(r)
        crb = combine_data(cl, device=dev)
        rrb = combine_data(rl, device=dev, other_with_same_sample_sizes=crb)
        scales.append((torch.empty((B, SH // s, SW // s), device=dev), crb, rrb))

    def step():
        for hm, c, r in scales:
            draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=True)

    t = _timeit(step, 50, 500, torch.cuda.synchronize)
    nbytes = sum(hm.numel() * 4 for hm, _, _ in scales)
    tg = _graph_time(step)
    print(json.dumps({"config": "C3", "metric": "multi-scale heat-maps strides 4/8/16 of 3840x2160, batch 32",
                      "ms_per_batch": t * 1e3, "frames_per_s": B / t, "GBps": nbytes / t / 1e9,
                      "bytes_per_frame": nbytes // B, "hipgraph_ms_per_batch": tg * 1e3,
                      "hipgraph_frames_per_s": B / tg, "hipgraph_GBps": nbytes / tg / 1e9}))

    # the same three maps from the FLOAT boxes in one launch (front end fused into the cull), vs front end + draw per scale
    from accvlab.draw_heatmap import draw_heatmap_multiscale, get_centers_and_radii

    crb_f = combine_data(centers_f, device=dev)
    brb_f = combine_data(boxes_f, device=dev, other_with_same_sample_sizes=crb_f)
    maps = [hm for hm, _, _ in scales]

    def step_unfused():
        for hm, s in zip(maps, (4.0, 8.0, 16.0)):
            ci, ri = get_centers_and_radii(crb_f, brb_f, s)
            draw_heatmap_batched(hm, ci, ri, 6.0, 1.0, clear=True)

    def step_fused():
        draw_heatmap_multiscale(maps, crb_f, brb_f, (4.0, 8.0, 16.0), 6.0, 1.0, clear=True)

    tu = _timeit(step_unfused, 50, 500, torch.cuda.synchronize)
    tf = _timeit(step_fused, 50, 500, torch.cuda.synchronize)
    print(json.dumps({"config": "C3 from float boxes", "metric": "bbox front end + 3 maps per batch of 32",
                      "per_scale_ops_ms": tu * 1e3, "draw_heatmap_multiscale_ms": tf * 1e3,
                      "multiscale_frames_per_s": B / tf, "multiscale_GBps": nbytes / tf / 1e9, "speedup": tu / tf}))

    # + lane raster: 8 lanes x 24 points per frame in source pixels, sampled at 256/128/64 arc-length positions (sample
    # spacing ~ the splat radius at every scale) and splatted with radius 2 into one lane map per scale
    # (sampler -> int targets -> fused clear+draw: 3 launches per scale)
    from accvlab.draw_heatmap import draw_polylines_batched

    L, P = 8, 24
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    lanes = torch.stack([xs, ys], -1).to(dev)
    lane_maps = [torch.empty_like(hm) for hm, _, _ in scales]

    def step_lanes():
        for (hm, c, r), lm, s in zip(scales, lane_maps, (4, 8, 16)):
            draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=True)
            draw_polylines_batched(lm, lanes, 1024 // s, 2, float(s), clear=True)

    t2 = _timeit(step_lanes, 50, 500, torch.cuda.synchronize)
    # everything fused: boxes of all scales in one launch + lanes of all scales in three (one sampling at 256 per lane)
    from accvlab.draw_heatmap import draw_polylines_multiscale

    def step_all_fused():
        draw_heatmap_multiscale(maps, crb_f, brb_f, (4.0, 8.0, 16.0), 6.0, 1.0, clear=True)
        draw_polylines_multiscale(lane_maps, lanes, 256, 2, (4.0, 8.0, 16.0), clear=True)

    t3 = _timeit(step_all_fused, 50, 500, torch.cuda.synchronize)
    t3l = _timeit(lambda: draw_polylines_multiscale(lane_maps, lanes, 256, 2, (4.0, 8.0, 16.0), clear=True), 50, 500,
                  torch.cuda.synchronize)
    print(json.dumps({"config": "C3+lanes fused", "metric": "float boxes + lanes (256 samples per lane at every scale), "
                      "4 launches per batch of 32", "ms_per_batch": t3 * 1e3, "frames_per_s": B / t3,
                      "GBps": 2 * nbytes / t3 / 1e9, "lane_raster_only_ms": t3l * 1e3}))
    print(json.dumps({"config": "C3+lanes", "metric": "C3 + lane raster (8 lanes x 24 pts, 256/128/64 samples, r=2) per scale",
                      "ms_per_batch": t2 * 1e3, "frames_per_s": B / t2, "GBps": 2 * nbytes / t2 / 1e9,
                      "bytes_per_frame": 2 * nbytes // B, "lane_part_ms": (t2 - t) * 1e3,
                      "hipgraph_ms_per_batch": (tg2 := _graph_time(step_lanes)) * 1e3,
                      "hipgraph_frames_per_s": B / tg2}))


def h2():
    """Ragged gather / compaction at StreamPETR-like shapes (batch 8, 900 queries, 256 channels, <= 100 targets):
    this build's kernels vs the torch-op formulations (per-sample python indexing, and the boolean-indexing
    formulation of batched_bool_indexing.py:195-221 in the reference)."""
    from accvlab.batching_helpers import RaggedBatch, batched_bool_indexing, batched_indexing_access

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    B, Q, D, K = 8, 900, 256, 100
    data = torch.randn(B, Q, D, generator=g).to(dev)
    sizes = torch.randint(1, K + 1, (B,), generator=g)
    idx = torch.stack([torch.randperm(Q, generator=g)[:K] for _ in range(B)]).to(dev)
    irb = RaggedBatch(idx, sample_sizes=sizes.to(dev))
    sizes_l = sizes.tolist()
    sync = torch.cuda.synchronize

    def loop_gather():
        out = torch.zeros(B, K, D, device=dev)
        for b in range(B):
            out[b, : sizes_l[b]] = data[b, idx[b, : sizes_l[b]]]
        return out

    t_k = _timeit(lambda: batched_indexing_access(data, irb, 0.0), 20, 200, sync)
    t_l = _timeit(loop_gather, 5, 50, sync)
    mask = (torch.rand(B, Q, generator=g) < 0.1).to(dev)

    def torch_bool_compaction():
        n = mask.sum(1)
        m = int(n.max().item())
        out = torch.zeros(B, m, D, device=dev)
        keep = torch.arange(m, device=dev).unsqueeze(0) < n.unsqueeze(1)
        out[keep] = data[mask]
        return out

    t_ck = _timeit(lambda: batched_bool_indexing(data, mask), 20, 200, sync)
    t_ct = _timeit(torch_bool_compaction, 10, 100, sync)
    print(json.dumps({"config": "H2", "shape": [B, Q, D, K], "ragged_gather_us": t_k * 1e6,
                      "per_sample_index_loop_us": t_l * 1e6, "gather_speedup": t_l / t_k,
                      "bool_compaction_us": t_ck * 1e6, "torch_boolean_indexing_us": t_ct * 1e6,
                      "compaction_speedup": t_ct / t_ck}))


def f3():
    """Loss-side caller pattern (examples/matched_loss.py): batched formulation over this build's ragged
    operators vs the per-sample loop, forward + backward, assignment (scipy, CPU) included in both and also timed
    without it."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import matched_loss as ml
    import accvlab.batching_helpers as bh

    dev = torch.device("cuda", 0)
    B, Q, C, G = 8, 900, 10, 100
    gb, gl, gw, pb, ps, pe = ml.make_inputs(B, Q, C, G, dev, seed=0, min_gt=1)
    sync = torch.cuda.synchronize

    def fwd_bwd(fn):
        leaves = [t.clone().requires_grad_(True) for t in (pb, ps, pe)]
        fn(gb, gl, gw, *leaves).sum().backward()

    t_b = _timeit(lambda: fwd_bwd(ml.run_batched), 3, 20, sync)
    t_l = _timeit(lambda: fwd_bwd(ml.loss_per_sample), 3, 20, sync)
    # loss only (matches precomputed): isolates the ragged gathers/scatters from the CPU assignment
    boxes = bh.combine_data(gb)
    labels = bh.combine_data(gl, other_with_same_sample_sizes=boxes)
    weights = bh.combine_data(gw, other_with_same_sample_sizes=boxes)
    m_gt, m_pred = ml.match_batched(boxes, labels, pb, ps)

    def loss_only():
        leaves = [t.clone().requires_grad_(True) for t in (pb, ps, pe)]
        ml.loss_batched(boxes, labels, weights, *leaves, m_gt, m_pred).sum().backward()

    t_lo = _timeit(loss_only, 5, 50, sync)
    # the fused kernel (accv_matched_pair_reduce_f32) vs the composition it replaces: matched L1 box loss, weighted,
    # summed per sample — 3 gathers + element-wise + masked sum (and their autograd nodes) vs ONE launch per direction
    gt_boxes = boxes.tensor.contiguous()
    w_t = weights.tensor.contiguous()

    def composition(pred):
        ga = bh.batched_indexing_access(gt_boxes, m_gt)
        gp = bh.batched_indexing_access(pred, m_pred)
        gw_ = bh.batched_indexing_access(w_t, m_gt)
        per_obj = (ga.tensor - gp.tensor).abs().sum(-1) * gw_.tensor
        return bh.sum_over_targets(ga.create_with_sample_sizes_like_self(per_obj, non_uniform_dim=1))

    def fused(pred):
        return bh.matched_pair_loss_sum(gt_boxes, pred, m_gt, m_pred, w_t, kind="l1")

    def fb(fn):
        p = pb.clone().requires_grad_(True)
        fn(p).sum().backward()

    with torch.no_grad():
        err = float((composition(pb) - fused(pb)).abs().max())
    t_cf = _timeit(lambda: composition(pb), 20, 200, sync)
    t_ff = _timeit(lambda: fused(pb), 20, 200, sync)
    t_cb = _timeit(lambda: fb(composition), 10, 100, sync)
    t_fb = _timeit(lambda: fb(fused), 10, 100, sync)
    print(json.dumps({"config": "F3 fused kernel", "shape": {"batch": B, "queries": Q, "max_gt": G, "box_dims": int(pb.shape[-1])},
                      "composition_fwd_us": t_cf * 1e6, "fused_fwd_us": t_ff * 1e6, "fwd_speedup": t_cf / t_ff,
                      "composition_fwd_bwd_us": t_cb * 1e6, "fused_fwd_bwd_us": t_fb * 1e6, "fwd_bwd_speedup": t_cb / t_fb,
                      "max_abs_difference": err}))
    # round 3: the reference example's own per-object losses through the fused op — box overlap (1 - IoU) and L1 between
    # one-hot labels and scores — against the composition the example writes (gathers + element-wise code + masked sum)
    lab_t = labels.tensor.contiguous()

    def comp_iou(pred):
        ga = bh.batched_indexing_access(gt_boxes, m_gt)
        gp = bh.batched_indexing_access(pred, m_pred)
        gw_ = bh.batched_indexing_access(w_t, m_gt)
        per_obj = (1.0 - ml._iou(ga.tensor, gp.tensor)) * gw_.tensor
        return bh.sum_over_targets(ga.create_with_sample_sizes_like_self(per_obj, non_uniform_dim=1))

    def fused_iou(pred):
        return bh.matched_pair_loss_sum(gt_boxes, pred, m_gt, m_pred, w_t, kind="iou_xyxy", eps=ml.EPS)

    def comp_cls(scores):
        gl_ = bh.batched_indexing_access(lab_t, m_gt)
        gs = bh.batched_indexing_access(scores, m_pred)
        gw_ = bh.batched_indexing_access(w_t, m_gt)
        per_obj = gw_.tensor * (gs.tensor - ml._one_hot(gl_.tensor.to(torch.int64), C)).abs().sum(-1)
        return bh.sum_over_targets(gl_.create_with_sample_sizes_like_self(per_obj, non_uniform_dim=1))

    def fused_cls(scores):
        return bh.matched_pair_loss_sum(lab_t, scores, m_gt, m_pred, w_t, kind="onehot_l1")

    def fb2(fn, x):
        p = x.clone().requires_grad_(True)
        fn(p).sum().backward()

    for name, comp, fus, x in (("iou_xyxy", comp_iou, fused_iou, pb), ("onehot_l1", comp_cls, fused_cls, ps)):
        with torch.no_grad():
            err2 = float((comp(x) - fus(x)).abs().max())
        print(json.dumps({"config": f"F3 fused kernel, kind {name}", "composition_fwd_us": _timeit(lambda: comp(x), 20, 200, sync) * 1e6,
                          "fused_fwd_us": _timeit(lambda: fus(x), 20, 200, sync) * 1e6,
                          "composition_fwd_bwd_us": _timeit(lambda: fb2(comp, x), 10, 100, sync) * 1e6,
                          "fused_fwd_bwd_us": _timeit(lambda: fb2(fus, x), 10, 100, sync) * 1e6, "max_abs_difference": err2}))
    for dt in (torch.float16, torch.bfloat16, torch.float64):
        a16, p16, w16 = gt_boxes.to(dt), pb.to(dt), w_t.to(dt)
        print(json.dumps({"config": f"F3 fused kernel, kind l1, {dt}",
                          "fused_fwd_us": _timeit(lambda: bh.matched_pair_loss_sum(a16, p16, m_gt, m_pred, w16, kind="l1"), 20, 200, sync) * 1e6}))

    def loss_only_fused():
        leaves = [t.clone().requires_grad_(True) for t in (pb, ps, pe)]
        ml.loss_batched_fused(boxes, labels, weights, *leaves, m_gt, m_pred).sum().backward()

    t_lof = _timeit(loss_only_fused, 5, 50, sync)
    print(json.dumps({"config": "F3", "shape": {"batch": B, "queries": Q, "classes": C, "max_gt": G},
                      "batched_fwd_bwd_ms": t_b * 1e3, "per_sample_loop_fwd_bwd_ms": t_l * 1e3,
                      "speedup": t_l / t_b, "batched_loss_only_fwd_bwd_ms": t_lo * 1e3,
                      "batched_loss_only_fused_class_and_box_terms_fwd_bwd_ms": t_lof * 1e3}))


class _MetaDataset(torch.utils.data.Dataset):
    """Each sample: 32 small fp32/int64 tensors (per-object meta data of one frame), like config C2's leaves."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(i)
        out = {"idx": i, "t": []}
        for j in range(32):
            k = int(torch.randint(1, 129, (1,), generator=g))
            out["t"].append(torch.randn(k, 4, generator=g) if j % 2 == 0 else torch.randint(0, 1000, (k,), generator=g))
        return out


def _identity_collate(samples):
    return samples


def f4():
    """DataLoader hook: batches of 64 samples x 32 small tensors (2048 tensors, ~2.4 MB) through a
    4-worker DataLoader with pin_memory, then onto the GPU with start_copy — default path (every tensor crosses the
    process boundary and is pinned on its own) vs packing_collate (one buffer)."""
    from accvlab.multi_tensor_copier import packing_collate, start_copy

    torch.multiprocessing.set_sharing_strategy("file_system")   # the plain path runs out of file descriptors otherwise
    dev = torch.device("cuda", 0)
    ds = _MetaDataset(64 * 24)

    def epoch(collate):
        loader = torch.utils.data.DataLoader(ds, batch_size=64, num_workers=4, pin_memory=True, collate_fn=collate,
                                             persistent_workers=False)
        it = iter(loader)
        first = next(it)                       # worker start-up is not what is measured
        start_copy(first, dev).get()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for batch in it:
            start_copy(batch, dev).get()
            n += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_plain = epoch(_identity_collate)
    t_packed = epoch(packing_collate())
    print(json.dumps({"config": "F4", "metric": "ms per batch of 2048 small tensors: 4-worker DataLoader(pin_memory) -> GPU",
                      "plain_ms": t_plain * 1e3, "packing_collate_ms": t_packed * 1e3, "speedup": t_plain / t_packed}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C0,C2,C3,H2,F3,F4")
    ap.add_argument("--tensors", type=int, default=10_000)
    a = ap.parse_args()
    which = a.configs.split(",")
    if "C0" in which:
        c0()
    if torch.cuda.is_available():
        if "C2" in which:
            c2(a.tensors)
            c2(528)
            c2_reverse(a.tensors)
        if "C3" in which:
            c3()
        if "H2" in which:
            h2()
        if "F3" in which:
            f3()
        if "F4" in which:
            f4()
