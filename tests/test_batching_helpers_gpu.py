"""GPU parity tests of the ragged-batch kernels (python operator API -> C-ABI -> HIP) against oracle/h2.py.

Bars (as in the reference's tests, packages/batching_helpers/tests/test_batched_indexing_ops.py:517,612,680,832
and :728,756,807): copies bit-exact (max_abs_diff == 0), accumulated values within 1e-6 (fp32/fp64) — the
tolerances for f16/bf16 accumulation are one ulp of the format and written at the test."""
import numpy as np
import pytest
import torch

import bench_workloads as wl
import h2_literals as lit
from oracle import h2 as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64]


def _bh():
    import accvlab.batching_helpers as bh
    return bh


def _np(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()
    if t.dtype == torch.float16:
        return t.numpy()
    return t.numpy()


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _rand_data(g, shape, dtype):
    if dtype in (torch.int32, torch.int64):
        return torch.from_numpy(g.randint(-1000, 1000, shape)).to(dtype)
    return torch.from_numpy(g.randn(*shape)).to(dtype)


def _cases(g, dtype, n_idx_dtype):
    """random (data [B,W,*inner], idx [B,K], counts [B]) with unique indices per row, some negative."""
    b, w, k = int(g.randint(1, 7)), int(g.randint(1, 40)), None
    inner = [(), (3,), (2, 5), (16,), (1,)][int(g.randint(0, 5))]
    k = int(g.randint(1, w + 1))
    data = _rand_data(g, (b, w) + inner, dtype)
    idx = np.stack([g.permutation(w)[:k] for _ in range(b)]).astype(np.int64)
    neg = g.rand(b, k) < 0.2
    idx = np.where(neg, idx - w, idx)
    counts = g.randint(0, k + 1, b).astype(np.int64)
    for i in range(b):
        idx[i, counts[i]:] = lit.BIG + w  # would fault if read
    return data, torch.from_numpy(idx).to(n_idx_dtype), torch.from_numpy(counts)


# ----------------------------------------------------------------------------------------- literals
@pytest.mark.parametrize("dtype", DTYPES)
def test_gather_literal_forward_and_backward(dtype):
    bh = _bh()
    data, idx, counts, fill, _, grad = lit.gather_literal(np.float64)
    fill_v = fill if dtype.is_floating_point else -4
    x = _t(data, dtype)
    rb = bh.RaggedBatch(_t(idx), sample_sizes=_t(counts))
    if dtype.is_floating_point:
        x.requires_grad_(True)
    out = bh.batched_indexing_access(x, rb, fill_v)
    assert isinstance(out, bh.RaggedBatch) and out.shape == (2, 3, 3, 2) and out.num_batch_dims == 2
    exp = oracle.gather(data.reshape(6, 5, 2), idx.reshape(6, 3), counts.reshape(6), fill_v).reshape(2, 3, 3, 2)
    assert torch.equal(out.tensor.detach().cpu(), torch.from_numpy(exp).to(dtype))
    if dtype in (torch.float32, torch.float64):
        torch.sin(out.tensor).sum().backward()  # gradient flows only through gathered entries
        assert torch.allclose(x.grad.cpu(), torch.from_numpy(grad).to(dtype), atol=1e-6)


def test_mask_from_indices_literal_and_multi_batch():
    bh = _bh()
    idx, counts, n, exp = lit.mask_from_indices_literal()
    m = bh.get_mask_from_indices(n, bh.RaggedBatch(_t(idx), sample_sizes=_t(counts)))
    assert m.dtype == torch.bool and np.array_equal(m.cpu().numpy(), exp)
    idx2 = np.stack([idx, idx[::-1]])
    c2 = np.stack([counts, counts[::-1]])
    m2 = bh.get_mask_from_indices(n, bh.RaggedBatch(_t(idx2), sample_sizes=_t(c2)))
    assert np.array_equal(m2.cpu().numpy(), np.stack([exp, exp[::-1]]))
    m32 = bh.get_mask_from_indices(n, bh.RaggedBatch(_t(idx, torch.int32), sample_sizes=_t(counts, torch.int32)))
    assert np.array_equal(m32.cpu().numpy(), exp)


def test_compaction_literal_on_gpu():
    bh = _bh()
    mask, datas, sizes, exps = lit.compaction_literal()
    out = bh.get_compact_lists(_t(mask), [_t(datas[0]), "x", _t(datas[1])])
    assert out[1] == "x"
    for rb, e in zip((out[0], out[2]), exps):
        assert np.array_equal(rb.tensor.cpu().numpy(), e) and np.array_equal(rb.sample_sizes.cpu().numpy(), sizes)
    idx = bh.get_indices_from_mask(_t(mask))
    eidx, esz = oracle.indices_from_mask(mask)
    assert idx.tensor.dtype == torch.int64
    assert np.array_equal(idx.tensor.cpu().numpy(), eidx) and np.array_equal(idx.sample_sizes.cpu().numpy(), esz)


@pytest.mark.parametrize("dtype", [torch.float32, torch.int64, torch.bfloat16])
def test_mapping_literal(dtype):
    bh = _bh()
    src, si, di, counts, into, exp = lit.mapping_literal()
    rb_s = bh.RaggedBatch(_t(si), sample_sizes=_t(counts))
    rb_d = bh.RaggedBatch(_t(di), sample_sizes=_t(counts))
    out = bh.batched_index_mapping(_t(src, dtype), rb_s, rb_d, _t(into, dtype))
    assert torch.equal(out.cpu(), torch.from_numpy(exp).to(dtype))


# ----------------------------------------------------------------------------------------- randomized copies
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_random_gather_scatter_insert_bit_exact(dtype, idx_dtype):
    bh = _bh()
    g = np.random.RandomState(DTYPES.index(dtype) * 2 + (idx_dtype == torch.int32))
    for trial in range(25):
        data, idx, counts = _cases(g, dtype, idx_dtype)
        cnt = counts.to(idx_dtype) if trial % 2 else counts
        rb = bh.RaggedBatch(idx.to(DEV), sample_sizes=cnt.to(DEV))
        d_np = data.float().numpy() if dtype in (torch.bfloat16, torch.float16) else data.numpy()
        # gather
        out = bh.batched_indexing_access(data.to(DEV), rb, 7.0)
        exp = oracle.gather(d_np, idx.numpy(), counts.numpy(), 7.0)
        assert torch.equal(out.tensor.cpu(), torch.from_numpy(exp).to(dtype)), f"gather trial {trial}"
        # inverse: scatter the gathered rows into a fresh tensor
        k = idx.shape[1]
        src = data[:, :k].contiguous()
        s_np = d_np[:, :k]
        w_out = data.shape[1]  # same width: negative indices wrap to the same (unique) targets
        inv = bh.batched_inverse_indexing_access(src.to(DEV), rb, w_out, -2.0)
        exp = oracle.scatter_new(s_np, idx.numpy(), counts.numpy(), w_out, -2.0, False)
        assert torch.equal(inv.cpu(), torch.from_numpy(exp).to(dtype)), f"inverse trial {trial}"
        # write into existing
        into = _rand_data(g, (data.shape[0], w_out) + tuple(data.shape[2:]), dtype)
        i_np = into.float().numpy() if dtype in (torch.bfloat16, torch.float16) else into.numpy()
        wr = bh.batched_indexing_write(src.to(DEV), rb, into.to(DEV))
        exp = oracle.scatter_insert(s_np, idx.numpy(), counts.numpy(), i_np)
        assert torch.equal(wr.cpu(), torch.from_numpy(exp).to(dtype)), f"write trial {trial}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64])
def test_accumulating_backward_with_duplicate_indices(dtype):
    """backward of the gather == accumulate-scatter; duplicates must add up (reference bar < 1e-6 for fp32/64)."""
    import accvlab.batching_helpers.batched_indexing_access_cuda as ext
    g = np.random.RandomState(11)
    for trial in range(15):
        b, w, k = int(g.randint(1, 5)), int(g.randint(1, 9)), int(g.randint(1, 30))
        inner = [(), (4,), (3, 2)][trial % 3]
        if dtype in (torch.int32, torch.int64):
            grad = torch.from_numpy(g.randint(-50, 50, (b, k) + inner)).to(dtype)
        else:
            grad = torch.from_numpy(g.randint(-8, 8, (b, k) + inner) / 4.0).to(dtype)  # exactly representable
        idx = torch.from_numpy(g.randint(0, w, (b, k)))
        counts = torch.from_numpy(g.randint(0, k + 1, b))
        for fill in (0.0, 3.0):
            out = ext.backward_new_tensor(grad.to(DEV), idx.to(DEV), counts.to(DEV), w, fill, backward_accumulate=True)
            gn = grad.double().numpy() if dtype.is_floating_point else grad.numpy()
            exp = oracle.scatter_new(gn, idx.numpy(), counts.numpy(), w, fill, True)
            # values are multiples of 1/4 with small magnitude: every partial sum is exact in all six dtypes
            assert torch.equal(out.cpu(), torch.from_numpy(exp).to(dtype)), f"trial {trial} fill {fill}"


def test_accumulate_float_random_tolerance():
    import accvlab.batching_helpers.batched_indexing_access_cuda as ext
    g = np.random.RandomState(5)
    grad = torch.from_numpy(g.randn(4, 200, 8)).float()
    idx = torch.from_numpy(g.randint(0, 10, (4, 200)))
    counts = torch.tensor([200, 150, 0, 77])
    out = ext.backward_new_tensor(grad.to(DEV), idx.to(DEV), counts.to(DEV), 10, 0.0, True)
    exp = oracle.scatter_new(grad.double().numpy(), idx.numpy(), counts.numpy(), 10, 0.0, True)
    assert np.abs(out.cpu().double().numpy() - exp).max() < 2e-5  # ~20 adds of |x|~1 per slot in fp32
    # bf16 / f16: one rounding per add
    for dt, tol in ((torch.float16, 5e-2), (torch.bfloat16, 4e-1)):
        o = ext.backward_new_tensor(grad.to(DEV, dt), idx.to(DEV), counts.to(DEV), 10, 0.0, True)
        assert np.abs(o.cpu().double().numpy() - exp).max() < tol


def test_index_mapping_random_and_backward():
    bh = _bh()
    g = np.random.RandomState(2)
    for trial in range(20):
        b, ws, wd, k = int(g.randint(1, 5)), int(g.randint(1, 12)), int(g.randint(1, 12)), None
        k = int(g.randint(1, wd + 1))
        inner = [(), (3,), (2, 2)][trial % 3]
        src = torch.from_numpy(g.randn(b, ws, *inner)).float()
        into = torch.from_numpy(g.randn(b, wd, *inner)).float()
        si = torch.from_numpy(g.randint(0, ws, (b, k)))
        di = torch.from_numpy(np.stack([g.permutation(wd)[:k] for _ in range(b)]))
        counts = torch.from_numpy(g.randint(0, k + 1, b))
        s, t = src.to(DEV).requires_grad_(True), into.to(DEV).requires_grad_(True)
        out = bh.batched_index_mapping(s, bh.RaggedBatch(si.to(DEV), sample_sizes=counts.to(DEV)),
                                       bh.RaggedBatch(di.to(DEV), sample_sizes=counts.to(DEV)), t)
        exp = oracle.map_pairs(src.numpy(), si.numpy(), di.numpy(), counts.numpy(), into.numpy())
        assert np.array_equal(out.detach().cpu().numpy(), exp)
        wgt = torch.from_numpy(g.randn(*out.shape)).float().to(DEV)
        (out * wgt).sum().backward()
        # d/dsrc: weights routed back along the pairs (sources may repeat -> accumulate); d/dinto: weights
        # except at overwritten slots
        gs = oracle.map_pairs(wgt.cpu().numpy(), di.numpy(), si.numpy(), counts.numpy(), np.zeros_like(src.numpy()), True)
        gt = oracle.insert_const(0.0, di.numpy(), counts.numpy(), wgt.cpu().numpy())
        assert np.abs(s.grad.cpu().numpy() - gs).max() < 1e-5
        assert np.array_equal(t.grad.cpu().numpy(), gt)


# ----------------------------------------------------------------------------------------- pad fill / mask
@pytest.mark.parametrize("dtype", DTYPES + [torch.bool])
def test_pad_fill_and_lazy_mask(dtype):
    bh = _bh()
    g = np.random.RandomState(4)
    for trial in range(10):
        b, w = int(g.randint(1, 9)), int(g.randint(1, 70))
        inner = [(), (5,), (2, 3)][trial % 3]
        data = (torch.from_numpy(g.rand(b, w, *inner)) > 0.5) if dtype == torch.bool else _rand_data(g, (b, w) + inner, dtype)
        sizes = torch.from_numpy(g.randint(0, w + 1, b))
        rb = bh.RaggedBatch(data.to(DEV), sample_sizes=sizes.to(DEV))
        val = True if dtype == torch.bool else (3 if not dtype.is_floating_point else -1.5)
        filled = rb.with_padded_set_to(val)
        d_np = data.float().numpy() if dtype in (torch.float16, torch.bfloat16) else data.numpy()
        exp = torch.from_numpy(oracle.pad_fill(d_np, sizes.numpy(), val)).to(dtype)
        assert torch.equal(filled.tensor.cpu(), exp)
        assert torch.equal(rb.tensor.cpu(), data)                       # out of place
        assert np.array_equal(rb.mask.cpu().numpy(), np.arange(w)[None] < sizes.numpy()[:, None])
        rb.set_padded_to(val)                                           # in place
        assert torch.equal(rb.tensor.cpu(), exp)


def test_multi_batch_dims_and_non_default_dim():
    bh = _bh()
    g = np.random.RandomState(9)
    data = torch.from_numpy(g.randn(2, 3, 4, 6)).float()       # batch (2,3), channels 4, width 6 at dim 3
    idx = torch.from_numpy(np.stack([g.permutation(6)[:5] for _ in range(6)]).reshape(2, 3, 5))
    counts = torch.from_numpy(g.randint(0, 6, (2, 3)))
    rb = bh.RaggedBatch(idx.to(DEV), sample_sizes=counts.to(DEV))
    out = bh.batched_indexing_access(data.to(DEV), rb, 0.0, dim_to_index_in=3)
    assert out.shape == (2, 3, 4, 5) and out.non_uniform_dim == 3
    exp = oracle.gather(data.permute(0, 1, 3, 2).reshape(6, 6, 4).numpy(), idx.reshape(6, 5).numpy(),
                        counts.reshape(6).numpy(), 0.0).reshape(2, 3, 5, 4).transpose(0, 1, 3, 2)
    assert np.array_equal(out.tensor.cpu().numpy(), exp)


# ----------------------------------------------------------------------------------------- compaction
def test_bool_indexing_random_gpu_vs_oracle():
    bh = _bh()
    g = np.random.RandomState(1)
    for trial in range(30):
        b, m = int(g.randint(1, 9)), int(g.randint(1, 300))
        inner = [(), (4,), (2, 3)][trial % 3]
        dtype = [torch.float32, torch.int64, torch.bool, torch.float16][trial % 4]
        data = (torch.from_numpy(g.rand(b, m, *inner)) > 0.5) if dtype == torch.bool else _rand_data(g, (b, m) + inner, dtype)
        mask = g.rand(b, m) > [0.5, 0.05, 0.95][trial % 3]
        valid = g.randint(0, m + 1, b)
        d_np = data.float().numpy() if dtype == torch.float16 else data.numpy()
        if trial % 2:
            got = bh.batched_bool_indexing(bh.RaggedBatch(data.to(DEV), sample_sizes=_t(valid)), _t(mask))
            exp, es = oracle.bool_compact(d_np, mask, valid)
        else:
            got = bh.batched_bool_indexing(data.to(DEV), _t(mask))
            exp, es = oracle.bool_compact(d_np, mask)
        assert np.array_equal(got.sample_sizes.cpu().numpy(), es)
        assert torch.equal(got.tensor.cpu(), torch.from_numpy(exp).to(dtype)), f"trial {trial}"
        # inverse write
        into = (torch.from_numpy(g.rand(b, m, *inner)) > 0.5) if dtype == torch.bool else _rand_data(g, (b, m) + inner, dtype)
        i_np = into.float().numpy() if dtype == torch.float16 else into.numpy()
        if trial % 2:
            back = bh.batched_bool_indexing_write(got, _t(mask), bh.RaggedBatch(into.to(DEV), sample_sizes=_t(valid)))
            eb = oracle.bool_write(exp, es, mask, i_np, valid)
            back = back.tensor
        else:
            back = bh.batched_bool_indexing_write(got, _t(mask), into.to(DEV))
            eb = oracle.bool_write(exp, es, mask, i_np)
        assert torch.equal(back.cpu(), torch.from_numpy(eb).to(dtype)), f"write trial {trial}"


def test_bool_indexing_differentiable():
    bh = _bh()
    x = torch.rand(3, 7, 2, device=DEV, requires_grad=True)
    mask = torch.rand(3, 7, device=DEV) > 0.4
    out = bh.batched_bool_indexing(x, mask)
    out.tensor.sum().backward()
    assert torch.equal(x.grad, mask.unsqueeze(-1).expand_as(x).float())


# ----------------------------------------------------------------------------------------- pack / split
def test_combine_data_device_pack_matches_cpu_and_split_roundtrip():
    bh = _bh()
    for dtype in (torch.float32, torch.int32, torch.float16, torch.int64):
        items = [(torch.rand(n, 4) * 100).to(dtype) for n in (3, 0, 17, 1, 128, 64)]
        cpu = bh.combine_data(items)
        for src in (items, [t.to(DEV) for t in items]):
            gpu = bh.combine_data(src, device=DEV)
            assert gpu.device.type == "cuda" and torch.equal(gpu.tensor.cpu(), cpu.tensor)
            assert torch.equal(gpu.sample_sizes.cpu(), cpu.sample_sizes)
            parts = gpu.split()
            assert all(torch.equal(p.cpu(), t) for p, t in zip(parts, items))
    # C1-style: centers (n,2) int32 and radii (n,) int32
    cl, rl = wl.heatmap_objects(64, 1080, 1920, 1, 128, "A", seed=42)
    c = bh.combine_data(cl, device=DEV)
    r = bh.combine_data(rl, device=DEV, other_with_same_sample_sizes=c)
    ec, es = oracle.combine([t.numpy() for t in cl])
    er, _ = oracle.combine([t.numpy() for t in rl])
    assert np.array_equal(c.tensor.cpu().numpy(), ec) and np.array_equal(r.tensor.cpu().numpy(), er)
    assert np.array_equal(c.sample_sizes.cpu().numpy(), es) and r.sample_sizes is not None
    a = torch.rand(3, 2, device=DEV, requires_grad=True)
    rb = bh.combine_data([a, torch.rand(5, 2, device=DEV)])
    rb.tensor.sum().backward()
    assert torch.all(a.grad == 1)


def test_error_behaviour_matches_reference():
    import accvlab.batching_helpers.batched_indexing_access_cuda as ext
    d = torch.zeros(2, 4, 3, device=DEV)
    i = torch.zeros(2, 2, dtype=torch.int64, device=DEV)
    n = torch.ones(2, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        ext.forward(d.transpose(1, 2), i, n)                 # not contiguous
    with pytest.raises(RuntimeError):
        ext.forward(d.cpu(), i, n)                           # not CUDA
    with pytest.raises(RuntimeError):
        ext.forward(d, i[:1], n)                             # batch mismatch
    with pytest.raises(RuntimeError):
        ext.backward_insert(d[:, :2].contiguous().double(), i, n, d)   # dtype mismatch
    with pytest.raises(RuntimeError):
        ext.map_values_by_index_pairs(d, i, i[:, :1].contiguous(), n, d)  # index shapes differ
    with pytest.raises(RuntimeError):
        ext.forward(d, i.float(), n)                         # non-integer indices
    # empty index tensor => early return with the filler (reference cu:253-255)
    out = ext.forward(d, torch.zeros(2, 0, dtype=torch.int64, device=DEV), n, 5.0)
    assert out.shape == (2, 0, 3)


@pytest.mark.parametrize("batch,width,k,inner,dtype", [
    (70_000, 3, 2, (1,), torch.float32),        # more samples than the grid's y extent: the sample loop of the row kernels
    (2, 5, 4, (4100,), torch.float32),          # 16 KB rows: more vectors per row than lanes per row (vector loop)
    (3, 7, 5, (3,), torch.float16),             # 6-byte rows: 2-byte vectors
    (3, 7, 5, (5,), torch.bool),                # 5-byte rows: byte vectors
    (1, 300, 300, (), torch.float64),           # scalar rows, one sample: 256 slots per workgroup
])
def test_row_kernels_launch_geometry_extremes(batch, width, k, inner, dtype):
    """the division-free 2-D launch geometry of the ragged byte movers (slot group x sample, 2^k lanes per row) at its
    corners, for gather (+fill), scatter, pair mapping, pad fill and pack, bit-exact against the oracle"""
    from accvlab.batching_helpers import batched_indexing_access_cuda as ext

    g = np.random.RandomState(batch % 1000 + width)
    shape = (batch, width) + inner
    if dtype.is_floating_point:
        data = torch.from_numpy(g.randn(*shape)).to(dtype)
    elif dtype == torch.bool:
        data = torch.from_numpy(g.randint(0, 2, shape)).to(dtype)
    else:
        data = torch.from_numpy(g.randint(0, 100, shape)).to(dtype)
    idx = np.stack([g.permutation(width)[:k] for _ in range(min(batch, 64))]).astype(np.int64)
    idx = np.tile(idx, (-(-batch // idx.shape[0]), 1))[:batch]
    counts = g.randint(0, k + 1, batch).astype(np.int64)
    d_dev, i_dev, c_dev = data.to(DEV), torch.from_numpy(idx).to(DEV), torch.from_numpy(counts).to(DEV)
    dn = data.numpy() if dtype != torch.bfloat16 else None
    # gather + fill through the public op's extension entry (float dtypes only there) or gather_rows
    out = torch.zeros((batch, k) + inner, dtype=dtype, device=DEV)
    ext.gather_rows(d_dev, i_dev, c_dev, k, out)
    want = oracle.gather(dn, idx, counts, 0)
    assert np.array_equal(out.cpu().numpy(), want)
    # scatter the gathered rows back into a marked copy
    into = torch.full(shape, 1 if dtype == torch.bool else 7, dtype=dtype)
    dst = into.to(DEV)
    ext.scatter_rows(out, i_dev, c_dev, k, dst)
    assert np.array_equal(dst.cpu().numpy(), oracle.scatter_insert(want, idx, counts, into.numpy()))
    # pad fill (SetPaddedTo) on a ragged view of the same data
    rb_counts = torch.from_numpy(g.randint(0, width + 1, batch).astype(np.int64))
    filled = d_dev.clone()
    ext.set_ragged_batch_padded_to_filler_value_in_place(filled, rb_counts.to(DEV), 3)
    assert np.array_equal(filled.cpu().numpy(), oracle.pad_fill(dn, rb_counts.numpy(), 3))
    # pack: flat rows -> padded
    if batch <= 1000:
        sizes = rb_counts
        flat = torch.cat([data[i, : int(sizes[i])] for i in range(batch)]) if int(sizes.sum()) else data[:0, 0]
        offs = torch.cumsum(sizes, 0) - sizes
        packed = ext.pack_rows(flat.to(DEV).contiguous(), offs.to(DEV), sizes.to(DEV), width)
        wantp = torch.zeros(shape, dtype=dtype)
        for i in range(batch):
            wantp[i, : int(sizes[i])] = data[i, : int(sizes[i])]
        assert torch.equal(packed.cpu(), wantp)
