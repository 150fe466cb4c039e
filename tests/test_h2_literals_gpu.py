"""The HIP path of batching_helpers against the reference-held known answers (tests/h2_literals.py: hand-written
expected tensors of packages/batching_helpers/tests/*, restated as data with file:line) — the same vectors that pin
oracle/h2.py on the CPU (tests/test_oracle_h2.py).  Bars as in the reference: copies bit-exact, gradients of exactly
representable inputs within 1e-6 (fp32/fp64), half formats within their rounding."""
import numpy as np
import pytest
import torch

import h2_literals as lit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}
FLOATS = [torch.float32, torch.float64, torch.float16, torch.bfloat16]
ALL = FLOATS + [torch.int32, torch.int64]


def _bh():
    import accvlab.batching_helpers as bh
    return bh


def _t(a, dtype=None, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    t = t.to(DEV)
    return t.requires_grad_(True) if grad else t


def _eq(got: torch.Tensor, want: np.ndarray, dtype):
    w = torch.from_numpy(np.ascontiguousarray(want)).to(dtype)
    g = got.detach().cpu()
    assert g.dtype == dtype and tuple(g.shape) == tuple(w.shape)
    assert torch.equal(torch.nan_to_num(g.double(), nan=-12345.0), torch.nan_to_num(w.double(), nan=-12345.0))


def _grad_close(got: torch.Tensor, want: np.ndarray, dtype):
    tol = {torch.float32: 1e-6, torch.float64: 1e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert float((got.detach().cpu().double() - torch.from_numpy(want).double()).abs().max()) <= tol


@pytest.mark.parametrize("dtype", ALL)
@pytest.mark.parametrize("fill", [0.0, 1.0, 2.3])
def test_inverse_indexing_literal(dtype, fill):
    bh = _bh()
    if not dtype.is_floating_point:
        fill = float(int(fill))
    data, idx, counts, n_targets, exp, grad = lit.inverse_literal(fill, np.float64)
    x = _t(data, dtype, grad=dtype.is_floating_point)
    rb = bh.RaggedBatch(_t(idx), sample_sizes=_t(counts))
    out = bh.batched_inverse_indexing_access(x, rb, n_targets, fill)
    # the filler is rounded to the tensor's dtype, as torch::full does (cpp:114)
    want = torch.from_numpy(exp).to(dtype)
    want[torch.from_numpy(exp == fill)] = torch.tensor(fill, dtype=dtype)
    assert torch.equal(out.detach().cpu(), want)
    if dtype.is_floating_point:
        torch.sin(out).sum().backward()
        # d/dx sum(sin(out)) at the scattered entries = cos(x); unused source slots get 0
        _grad_close(x.grad, grad, dtype)


@pytest.mark.parametrize("dtype", ALL)
def test_indexing_write_literal(dtype):
    bh = _bh()
    data, idx, counts, into, exp, g_src, g_into = lit.write_literal(np.float64)
    fl = dtype.is_floating_point
    x, y = _t(data, dtype, grad=fl), _t(into, dtype, grad=fl)
    rb = bh.RaggedBatch(_t(idx), sample_sizes=_t(counts))
    out = bh.batched_indexing_write(x, rb, y)
    _eq(out, exp, dtype)
    assert float(y.detach().cpu().double().max()) < 0           # out of place: the destination is untouched
    if fl:
        torch.sin(out).sum().backward()
        _grad_close(x.grad, g_src, dtype)
        _grad_close(y.grad, g_into, dtype)
    # RaggedBatch destination -> RaggedBatch result (batched_indexing_ops.py:363-455)
    sizes = _t(np.full((2, 3), 5, dtype=np.int64))
    out_rb = bh.batched_indexing_write(x.detach(), rb, bh.RaggedBatch(y.detach(), sample_sizes=sizes))
    assert isinstance(out_rb, bh.RaggedBatch)
    _eq(out_rb.tensor, exp, dtype)


@pytest.mark.parametrize("dtype", ALL + [torch.bool])
@pytest.mark.parametrize("multi", [False, True])
def test_bool_indexing_and_write_literals(dtype, multi):
    bh = _bh()
    data, sizes, mask, exp, out_sizes = (lit.bool_index_multi_literal if multi else lit.bool_index_simple_literal)(np.float64)
    if dtype == torch.bool:
        data, exp = (data % 3 == 0), None
    rb = bh.RaggedBatch(_t(data, dtype), sample_sizes=_t(sizes))
    mrb = bh.RaggedBatch(_t(mask), sample_sizes=_t(sizes))
    got = bh.batched_bool_indexing(rb, mrb)
    assert torch.equal(got.sample_sizes.cpu(), torch.from_numpy(out_sizes))
    if exp is not None:
        _eq(got.tensor, exp, dtype)
    # inverse: write the compacted rows back through the mask into a marked copy -> the original data
    if dtype == torch.bool:
        return
    compact, csz, mask, sizes, into, want = lit.bool_write_literal(multi, np.float64)
    if not dtype.is_floating_point:
        into = np.nan_to_num(into, nan=-999)
    back = bh.batched_bool_indexing_write(bh.RaggedBatch(_t(compact, dtype), sample_sizes=_t(csz)),
                                          bh.RaggedBatch(_t(mask), sample_sizes=_t(sizes)),
                                          bh.RaggedBatch(_t(into, dtype), sample_sizes=_t(sizes)))
    _eq(back.tensor, want, dtype)


@pytest.mark.parametrize("dtype", ALL)
@pytest.mark.parametrize("multi", [False, True])
def test_pad_fill_literal(dtype, multi):
    bh = _bh()
    data, sizes, value, exp, grad = lit.pad_fill_literal(multi, np.float64)
    fl = dtype.is_floating_point
    x = _t(data, dtype, grad=fl)
    rb = bh.RaggedBatch(x, sample_sizes=_t(sizes))
    out = rb.with_padded_set_to(value)
    _eq(out.tensor, exp, dtype)
    _eq(x, data, dtype)                                          # with_padded_set_to is out of place
    if fl:
        torch.sin(out.tensor).sum().backward()
        _grad_close(x.grad, grad, dtype)
    # in-place variant (ragged_batch.py:524-558)
    rb2 = bh.RaggedBatch(_t(data, dtype), sample_sizes=_t(sizes))
    rb2.set_padded_to(value)
    _eq(rb2.tensor, exp, dtype)


def test_indices_from_mask_literals():
    bh = _bh()
    (m1, _, rows1, w1), (m2, s2, rows2, w2) = lit.indices_from_mask_literals()
    i1 = bh.get_indices_from_mask(_t(m1))
    assert i1.tensor.dtype == torch.int64 and tuple(i1.tensor.shape) == (4, w1)
    i2 = bh.get_indices_from_mask(bh.RaggedBatch(_t(m2), sample_sizes=_t(s2)))
    assert i2.tensor.dtype == torch.int64 and tuple(i2.tensor.shape) == (3, w2)
    for got, rows in ((i1, rows1), (i2, rows2)):
        assert got.sample_sizes.cpu().tolist() == [len(r) for r in rows]
        for i, r in enumerate(rows):
            assert got.tensor[i, :len(r)].cpu().tolist() == r
    with pytest.raises(AssertionError):
        bh.get_indices_from_mask(torch.zeros(2, 3, 4, dtype=torch.bool, device=DEV))   # 2-D masks only (:577-628)


def test_combine_and_split_literal_shapes_on_gpu():
    bh = _bh()
    sh = lit.combine_literal_shapes()
    g = torch.Generator().manual_seed(0)
    flat = [torch.randn(n, generator=g).to(DEV) for n in sh["flat"]]
    c = bh.combine_data(flat)
    assert tuple(c.tensor.shape) == (3, 4) and c.tensor.device.type == "cuda"
    for i, n in enumerate(sh["flat"]):
        assert torch.equal(c.tensor[i, :n], flat[i])
    cn = bh.combine_data([[flat[0], flat[1]], [flat[2]]])
    assert torch.equal(cn.tensor, c.tensor)
    extra = [torch.randn(s, generator=g).to(DEV) for s in sh["extra"]]
    ce = bh.combine_data(extra)
    assert tuple(ce.tensor.shape) == (2, 3, 4) and torch.equal(ce.tensor[0, :2], extra[0]) and torch.equal(ce.tensor[1], extra[1])
    grid = [[torch.randn((n,) + sh["grid_inner"], generator=g).to(DEV) for n in row] for row in sh["grid"]]
    cg = bh.combine_data(grid, flatten_batch_dims=False)
    assert tuple(cg.batch_shape) == (2, 3) and tuple(cg.shape) == (2, 3, 7, 3, 4)
    parts = cg.split()
    for i, row in enumerate(sh["grid"]):
        for j, n in enumerate(row):
            assert int(cg.sample_sizes[i, j]) == n and torch.equal(cg.tensor[i, j, :n], grid[i][j])
            assert torch.equal(parts[i][j], grid[i][j])          # test_ragged_batch_functions.py:467-494
    with pytest.raises(AssertionError):
        bh.combine_data([])
    with pytest.raises(AssertionError):
        bh.combine_data([flat[0], "not a tensor", flat[1]])
    with pytest.raises(AssertionError):
        bh.combine_data([grid[0], grid[1] + [torch.randn(3, 3, 4, device=DEV)]], flatten_batch_dims=False)


@pytest.mark.parametrize("dtype", FLOATS)
def test_mapping_gradients_and_multi_batch_literal(dtype):
    bh = _bh()
    src, si, di, counts, into, exp = lit.mapping_literal(np.float64)
    g_src, g_into = lit.mapping_grads_literal(np.float64)
    x, y = _t(src, dtype, grad=True), _t(into, dtype, grad=True)
    out = bh.batched_index_mapping(x, bh.RaggedBatch(_t(si), sample_sizes=_t(counts)),
                                   bh.RaggedBatch(_t(di), sample_sizes=_t(counts)), y)
    _eq(out, exp, dtype)
    torch.sin(out).sum().backward()
    _grad_close(x.grad, g_src, dtype)
    _grad_close(y.grad, g_into, dtype)
    src, si, di, counts, into, exp, g_src, g_into = lit.mapping_multi_batch_literal(np.float64)
    x, y = _t(src, dtype, grad=True), _t(into, dtype, grad=True)
    out = bh.batched_index_mapping(x, bh.RaggedBatch(_t(si), sample_sizes=_t(counts)),
                                   bh.RaggedBatch(_t(di), sample_sizes=_t(counts)), y)
    _eq(out, exp, dtype)
    torch.sin(out).sum().backward()
    _grad_close(x.grad, g_src, dtype)
    _grad_close(y.grad, g_into, dtype)
