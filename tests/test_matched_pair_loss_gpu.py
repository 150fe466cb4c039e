"""F3 fused kernel (accv_matched_pair_reduce_f32 / _bwd_f32): parity against the COMPOSITION of the two pinned oracle
functions it fuses — oracle.h2.gather (pinned by the reference's gather literals) on both sides of the matching, the
element-wise loss in float64 numpy, and the masked per-sample sum (sum_over_targets semantics: valid entries only) —
and against the same composition written with this package's own operators (forward and gradients).
Tolerance: fp32 sums of <= 100 x 10 terms in a different order: 1e-5 relative."""
import numpy as np
import pytest
import torch

from oracle import h2 as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _loss_np(d, kind, beta):
    ad = np.abs(d)
    if kind == "l1":
        return ad
    if kind == "l2":
        return d * d
    return np.where(ad < beta, 0.5 * d * d / beta, ad - 0.5 * beta)


def _case(seed, b=8, na=100, nb=900, k=100, inner=(10,), idx_dtype=torch.int64):
    g = np.random.default_rng(seed)
    a = g.normal(size=(b, na) + inner).astype(np.float32)
    bb = g.normal(size=(b, nb) + inner).astype(np.float32)
    w = g.uniform(0.2, 2.0, size=(b, na)).astype(np.float32)
    counts = g.integers(0, min(k, na, nb) + 1, size=b).astype(np.int64)
    counts[0] = min(k, na, nb)
    ia = np.full((b, k), 10 ** 6, dtype=np.int64)
    ib = np.full((b, k), 10 ** 6, dtype=np.int64)
    for i in range(b):
        n = int(counts[i])
        ia[i, :n] = g.permutation(na)[:n]
        ib[i, :n] = g.permutation(nb)[:n]
        neg = g.random(n) < 0.2
        ib[i, :n][neg] -= nb                                          # negative indices wrap once
    return a, bb, w, ia, ib, counts, idx_dtype


@pytest.mark.parametrize("kind,beta", [("l1", 1.0), ("l2", 1.0), ("smooth_l1", 0.5)])
@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("shape", [dict(), dict(b=3, na=7, nb=5, k=6, inner=(2, 3), idx_dtype=torch.int32),
                                   dict(b=1, na=300, nb=300, k=300, inner=())])
def test_fused_matches_oracle_composition_and_operator_composition(kind, beta, weighted, shape):
    import accvlab.batching_helpers as bh

    a, b, w, ia, ib, counts, idt = _case(3, **shape)
    B, K = ia.shape
    # ---- oracle composition (float64)
    ga = oracle.gather(a.astype(np.float64), ia, counts, 0.0)
    gb = oracle.gather(b.astype(np.float64), ib, counts, 0.0)
    gw = oracle.gather(w.astype(np.float64), ia, counts, 0.0) if weighted else np.ones((B, K))
    per_obj = _loss_np(ga - gb, kind, beta).reshape(B, K, -1).sum(-1) * gw
    valid = np.arange(K)[None, :] < counts[:, None]
    want = (per_obj * valid).sum(1)
    # ---- fused op
    ta = torch.from_numpy(a).to(DEV).requires_grad_(True)
    tb = torch.from_numpy(b).to(DEV).requires_grad_(True)
    tw = torch.from_numpy(w).to(DEV).requires_grad_(True) if weighted else None
    ra = bh.RaggedBatch(torch.from_numpy(ia).to(idt).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    rb = bh.RaggedBatch(torch.from_numpy(ib).to(idt).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    out = bh.matched_pair_loss_sum(ta, tb, ra, rb, tw, kind=kind, beta=beta)
    assert out.shape == (B,) and out.dtype == torch.float32
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(out.detach().cpu().numpy() - want).max()) <= 1e-5 * scale
    up = torch.linspace(0.5, 1.5, B, device=DEV)
    (out * up).sum().backward()
    # ---- the same composition from this package's operators (5 gathers + element-wise + masked sums), autograd reference
    ca = torch.from_numpy(a).to(DEV).requires_grad_(True)
    cb = torch.from_numpy(b).to(DEV).requires_grad_(True)
    cw = torch.from_numpy(w).to(DEV).requires_grad_(True) if weighted else None
    g1 = bh.batched_indexing_access(ca, ra)
    g2 = bh.batched_indexing_access(cb, rb)
    d = g1.tensor - g2.tensor
    if kind == "l1":
        l = d.abs()
    elif kind == "l2":
        l = d * d
    else:
        l = torch.nn.functional.smooth_l1_loss(g1.tensor, g2.tensor, beta=beta, reduction="none")
    l = l.flatten(2).sum(-1) if l.dim() > 2 else l
    if weighted:
        l = l * bh.batched_indexing_access(cw, ra).tensor
    ref = bh.sum_over_targets(g1.create_with_sample_sizes_like_self(l, non_uniform_dim=1))
    assert float((out.detach() - ref.detach()).abs().max()) <= 1e-5 * scale
    (ref * up).sum().backward()
    for got, exp in ((ta.grad, ca.grad), (tb.grad, cb.grad)) + (((tw.grad, cw.grad),) if weighted else ()):
        assert float((got - exp).abs().max()) <= 1e-5 * max(1.0, float(exp.abs().max()))
    # run-to-run determinism of the forward (fixed-order reduction, no atomics)
    again = bh.matched_pair_loss_sum(ta.detach(), tb.detach(), ra, rb, tw.detach() if weighted else None, kind=kind, beta=beta)
    assert torch.equal(again, out.detach())


def test_fused_edge_cases_and_validation():
    import accvlab.batching_helpers as bh

    a = torch.rand(2, 4, 3, device=DEV)
    b = torch.rand(2, 5, 3, device=DEV)
    idx = torch.tensor([[0, 1, 2], [3, 9, 0]], device=DEV)            # 9 is out of range on both sides -> skipped
    ra = bh.RaggedBatch(idx, sample_sizes=torch.tensor([0, 3], device=DEV))
    out = bh.matched_pair_loss_sum(a, b, ra, ra)
    assert float(out[0]) == 0.0                                        # no matches -> 0, not NaN
    want = (a[1, 3] - b[1, 3]).abs().sum() + (a[1, 0] - b[1, 0]).abs().sum()
    assert abs(float(out[1]) - float(want)) <= 1e-6
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a.double(), b, ra, ra)                  # one dtype on both sides
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a.to(torch.int32), b.to(torch.int32), ra, ra)
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b, ra, ra, kind="iou_xyxy")          # rows of 3 are not boxes
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b, ra, ra, kind="onehot_l1")         # labels must be integers [B, N_a]
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b, ra, ra, kind="huber")
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a.cpu(), b, ra, ra)
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b[:, :, :2].contiguous(), ra, ra)
    e = bh.matched_pair_loss_sum(a[:0], b[:0], bh.RaggedBatch(idx[:0], sample_sizes=torch.zeros(0, dtype=torch.int64, device=DEV)),
                                 bh.RaggedBatch(idx[:0], sample_sizes=torch.zeros(0, dtype=torch.int64, device=DEV)))
    assert e.shape == (0,)


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the reference example's own per-object losses (packages/batching_helpers/example/loss_computation.py:225-274)
# and every dtype the replaced gathers accept.  Forward oracle = oracle.h2.gather (pinned) + the loss in float64 numpy;
# gradients = torch autograd over the same formulation as the example writes it (masked assignments, torch.max / torch.min).
def _iou_loss_np(g, p, eps):
    area_g = np.prod(g[..., 2:4] - g[..., 0:2], axis=-1)
    area_p = np.prod(p[..., 2:4] - p[..., 0:2], axis=-1)
    size = np.minimum(g[..., 2:4], p[..., 2:4]) - np.maximum(g[..., 0:2], p[..., 0:2])
    size = np.where(size < 0.0, 0.0, size)
    inter = np.prod(size, axis=-1)
    union = area_g + area_p - inter
    union = np.where(union < eps, eps, union)
    return 1.0 - inter / union


def _iou_loss_torch(g, p, eps):       # the example's formulation, operation by operation
    areas_g = torch.prod(g[..., 2:4] - g[..., 0:2], axis=-1)
    areas_p = torch.prod(p[..., 2:4] - p[..., 0:2], axis=-1)
    size = torch.min(g[..., 2:4], p[..., 2:4]) - torch.max(g[..., 0:2], p[..., 0:2])
    size = size.clone()
    size[size < 0.0] = 0.0
    inter = torch.prod(size, axis=-1)
    union = areas_g + areas_p - inter
    union = union.clone()
    union[union < eps] = eps
    return 1.0 - inter / union


def _box_case(seed, b, na, nb, k, ties=False):
    g = np.random.default_rng(seed)

    def boxes(n):
        tl = g.uniform(0, 60, size=(b, n, 2))
        return np.concatenate([tl, tl + g.uniform(2, 40, size=(b, n, 2))], -1).astype(np.float32)

    a, bb = boxes(na), boxes(nb)
    if ties:          # exact ties on single coordinates (max / min split the gradient), disjoint and degenerate boxes
        a, bb = np.round(a / 8) * 8, np.round(bb / 8) * 8
        a[:, 0] = [0, 0, 0, 0]                       # zero-area ground truth, union below eps against ...
        bb[:, 0] = [5, 5, 5, 5]                      # ... a zero-area prediction
    w = g.uniform(0.2, 2.0, size=(b, na)).astype(np.float32)
    counts = g.integers(0, min(k, na, nb) + 1, size=b).astype(np.int64)
    counts[0] = min(k, na, nb)
    ia = np.zeros((b, k), dtype=np.int64)
    ib = np.zeros((b, k), dtype=np.int64)
    for i in range(b):
        n = int(counts[i])
        ia[i, :n] = g.permutation(na)[:n]
        ib[i, :n] = g.permutation(nb)[:n]
        ia[i, n:] = 10 ** 6
        ib[i, n:] = 10 ** 6
    if ties:
        ia[0, 0], ib[0, 0] = 0, 0
    return a, bb, w, ia, ib, counts


_TOL = {torch.float32: 1e-5, torch.float64: 1e-12, torch.float16: 2e-3, torch.bfloat16: 2e-2}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ties", [False, True], ids=["random", "ties-and-degenerate"])
@pytest.mark.parametrize("weighted", [True, False])
def test_iou_kind_matches_oracle_composition_and_autograd(dtype, ties, weighted):
    import accvlab.batching_helpers as bh

    eps = 1e-6
    a, b, w, ia, ib, counts = _box_case(5, 4, 30, 50, 20, ties)
    B, K = ia.shape
    ta, tb, tw = (torch.from_numpy(x).to(DEV).to(dtype) for x in (a, b, w))
    # the oracle sees the values the kernel sees (the rounding to the data dtype is not the kernel's arithmetic)
    a64, b64, w64 = (t.to(torch.float64).cpu().numpy() for t in (ta, tb, tw))
    ga, gb = oracle.gather(a64, ia, counts, 0.0), oracle.gather(b64, ib, counts, 0.0)
    gw = oracle.gather(w64, ia, counts, 0.0) if weighted else np.ones((B, K))
    valid = np.arange(K)[None, :] < counts[:, None]
    want = (np.where(valid, _iou_loss_np(ga, gb, eps) * gw, 0.0)).sum(1)
    ta.requires_grad_(True), tb.requires_grad_(True), tw.requires_grad_(True)
    ra = bh.RaggedBatch(torch.from_numpy(ia).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    rb = bh.RaggedBatch(torch.from_numpy(ib).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    out = bh.matched_pair_loss_sum(ta, tb, ra, rb, tw if weighted else None, kind="iou_xyxy", eps=eps)
    assert out.dtype == (torch.float64 if dtype == torch.float64 else torch.float32) and out.shape == (B,)
    # forward: float32 (float64) arithmetic on the dtype-rounded inputs
    fwd_tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert float(np.abs(out.detach().cpu().numpy() - want).max()) <= fwd_tol * max(1.0, float(np.abs(want).max()))
    up = torch.linspace(0.5, 1.5, B, device=DEV, dtype=out.dtype)
    (out * up).sum().backward()
    # gradients: autograd over the example's formulation in float64 on the same (rounded) inputs
    ca, cb, cw = (torch.from_numpy(x).to(DEV).requires_grad_(True) for x in (a64, b64, w64))
    idx_a = torch.from_numpy(np.where(valid, ia, 0)).to(DEV)
    idx_b = torch.from_numpy(np.where(valid, ib, 0)).to(DEV)
    g1 = torch.gather(ca, 1, idx_a.unsqueeze(-1).expand(-1, -1, 4))
    g2 = torch.gather(cb, 1, idx_b.unsqueeze(-1).expand(-1, -1, 4))
    l = _iou_loss_torch(g1, g2, eps)
    if weighted:
        l = l * torch.gather(cw, 1, idx_a)
    ref = (l * torch.from_numpy(valid).to(DEV)).sum(1)
    (ref * up.double()).sum().backward()
    tol = _TOL[dtype]
    pairs = [(ta.grad, ca.grad), (tb.grad, cb.grad)] + ([(tw.grad, cw.grad)] if weighted else [])
    for got, exp in pairs:
        assert got.dtype == dtype
        assert float((got.double() - exp).abs().max()) <= tol * max(1.0, float(exp.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("label_dtype", [torch.int64, torch.int32])
def test_onehot_l1_kind_matches_the_example_formulation(dtype, label_dtype):
    import accvlab.batching_helpers as bh

    g = np.random.default_rng(9)
    B, NA, NB, K, C = 5, 12, 40, 10, 7
    labels = g.integers(0, C, size=(B, NA))
    labels[0, 0] = C + 3                                # outside [0, C): matches no class (one-hot row of zeros)
    scores = g.uniform(0, 1, size=(B, NB, C)).astype(np.float32)
    scores[0, 0, :] = [0, 1, 0, 1, 0.5, 0, 1]           # exact zeros of the difference: |.|' = 0 there
    labels[0, 1] = 1
    w = g.uniform(0.2, 2.0, size=(B, NA)).astype(np.float32)
    counts = g.integers(0, K + 1, size=B).astype(np.int64)
    counts[0] = K
    ia = np.full((B, K), 10 ** 6, dtype=np.int64)
    ib = np.full((B, K), 10 ** 6, dtype=np.int64)
    for i in range(B):
        n = int(counts[i])
        ia[i, :n] = g.permutation(NA)[:n]
        ib[i, :n] = g.permutation(NB)[:n]
    ia[0, :2], ib[0, :2] = [0, 1], [5, 0]
    ts = torch.from_numpy(scores).to(DEV).to(dtype)
    tw = torch.from_numpy(w).to(DEV).to(dtype)
    s64, w64 = ts.double().cpu().numpy(), tw.double().cpu().numpy()
    valid = np.arange(K)[None, :] < counts[:, None]
    glab = oracle.gather(labels, ia, counts, 0)
    gs = oracle.gather(s64, ib, counts, 0.0)
    gw = oracle.gather(w64, ia, counts, 0.0)
    onehot = (np.arange(C)[None, None, :] == glab[..., None]).astype(np.float64)
    want = np.where(valid, np.abs(onehot - gs).sum(-1) * gw, 0.0).sum(1)
    ts.requires_grad_(True), tw.requires_grad_(True)
    ra = bh.RaggedBatch(torch.from_numpy(ia).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    rb = bh.RaggedBatch(torch.from_numpy(ib).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    tl = torch.from_numpy(labels).to(label_dtype).to(DEV)
    out = bh.matched_pair_loss_sum(tl, ts, ra, rb, tw, kind="onehot_l1")
    fwd_tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert float(np.abs(out.detach().cpu().numpy() - want).max()) <= fwd_tol * max(1.0, float(np.abs(want).max()))
    out.sum().backward()
    cs, cw = (torch.from_numpy(x).to(DEV).requires_grad_(True) for x in (s64, w64))
    idx_a = torch.from_numpy(np.where(valid, ia, 0)).to(DEV)
    idx_b = torch.from_numpy(np.where(valid, ib, 0)).to(DEV)
    diff = torch.abs(torch.from_numpy(onehot).to(DEV) - torch.gather(cs, 1, idx_b.unsqueeze(-1).expand(-1, -1, C)))
    ref = ((torch.gather(cw, 1, idx_a).unsqueeze(-1) * diff).sum(2) * torch.from_numpy(valid).to(DEV)).sum(1)
    ref.sum().backward()
    tol = _TOL[dtype]
    for got, exp in ((ts.grad, cs.grad), (tw.grad, cw.grad)):
        assert got.dtype == dtype and float((got.double() - exp).abs().max()) <= tol * max(1.0, float(exp.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kind,beta", [("l1", 1.0), ("l2", 1.0), ("smooth_l1", 0.5)])
def test_elementwise_kinds_in_every_gather_dtype(dtype, kind, beta):
    import accvlab.batching_helpers as bh

    a, b, w, ia, ib, counts, _ = _case(4, b=3, na=20, nb=30, k=12, inner=(5,))
    B, K = ia.shape
    ta, tb, tw = (torch.from_numpy(x).to(DEV).to(dtype) for x in (a, b, w))
    a64, b64, w64 = (t.double().cpu().numpy() for t in (ta, tb, tw))
    ga, gb, gw = oracle.gather(a64, ia, counts, 0.0), oracle.gather(b64, ib, counts, 0.0), oracle.gather(w64, ia, counts, 0.0)
    valid = np.arange(K)[None, :] < counts[:, None]
    want = (_loss_np(ga - gb, kind, beta).sum(-1) * gw * valid).sum(1)
    ta.requires_grad_(True), tb.requires_grad_(True)
    ra = bh.RaggedBatch(torch.from_numpy(ia).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    rb = bh.RaggedBatch(torch.from_numpy(ib).to(DEV), sample_sizes=torch.from_numpy(counts).to(DEV))
    out = bh.matched_pair_loss_sum(ta, tb, ra, rb, tw, kind=kind, beta=beta)
    fwd_tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert float(np.abs(out.detach().cpu().numpy() - want).max()) <= fwd_tol * max(1.0, float(np.abs(want).max()))
    out.sum().backward()
    assert ta.grad.dtype == dtype and tb.grad.dtype == dtype
    assert torch.equal(ta.grad != 0, ta.grad != 0) and float(ta.grad.abs().sum()) > 0


def test_only_the_sample_sizes_of_indices_a_are_read():
    """documented semantics: slot j of sample i is a pair iff j < indices_a.sample_sizes[i]"""
    import accvlab.batching_helpers as bh

    a = torch.rand(2, 4, 3, device=DEV)
    b = torch.rand(2, 5, 3, device=DEV)
    idx = torch.tensor([[0, 1, 2], [3, 2, 0]], device=DEV)
    ra = bh.RaggedBatch(idx, sample_sizes=torch.tensor([2, 3], device=DEV))
    rb_other = bh.RaggedBatch(idx.clone(), sample_sizes=torch.tensor([1, 1], device=DEV))
    assert torch.equal(bh.matched_pair_loss_sum(a, b, ra, rb_other), bh.matched_pair_loss_sum(a, b, ra, ra))
