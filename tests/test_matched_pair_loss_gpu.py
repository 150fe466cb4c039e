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
        bh.matched_pair_loss_sum(a.double(), b.double(), ra, ra)
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b, ra, ra, kind="huber")
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a.cpu(), b, ra, ra)
    with pytest.raises(RuntimeError):
        bh.matched_pair_loss_sum(a, b[:, :, :2].contiguous(), ra, ra)
    e = bh.matched_pair_loss_sum(a[:0], b[:0], bh.RaggedBatch(idx[:0], sample_sizes=torch.zeros(0, dtype=torch.int64, device=DEV)),
                                 bh.RaggedBatch(idx[:0], sample_sizes=torch.zeros(0, dtype=torch.int64, device=DEV)))
    assert e.shape == (0,)
