"""Further behaviours the reference's batching_helpers tests pin (names of the reference tests in brackets): pad-fill
backward, multi-batch-dim / transposed boolean indexing and index mapping, gradient propagation through the scatter
ops, RaggedBatch conveniences, error paths.  Device-parametrised: CPU always, cuda:0 when marked gpu."""
import numpy as np
import pytest
import torch

from accvlab.batching_helpers import (RaggedBatch, batched_bool_indexing, batched_bool_indexing_write,
                                      batched_index_mapping, batched_indexing_access, batched_indexing_write,
                                      batched_inverse_indexing_access, get_mask_from_indices)
from oracle import h2 as oracle

CPU_AND_GPU = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]
GPU_ONLY = [pytest.param("cuda:0", marks=pytest.mark.gpu)]


# [test_ragged_batch_set_padded_to_backward(_example)(_multi_batch_dim)]
@pytest.mark.parametrize("dev", CPU_AND_GPU)
@pytest.mark.parametrize("batch_shape", [(4,), (2, 3)])
def test_set_padded_to_backward_zeroes_padding_gradient(dev, batch_shape):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(*batch_shape, 5, 3, generator=g).to(dev).requires_grad_(True)
    sizes = torch.randint(0, 6, batch_shape, generator=g).to(dev)
    rb = RaggedBatch(x, sample_sizes=sizes)
    y = rb.with_padded_set_to(7.0)
    w = torch.rand(*batch_shape, 5, 3, generator=g).to(dev)
    (y.tensor * w).sum().backward()
    valid = (torch.arange(5, device=dev).view(*([1] * len(batch_shape)), 5) < sizes.unsqueeze(-1)).unsqueeze(-1)
    assert torch.equal(x.grad, w * valid)
    assert torch.all(y.tensor[~valid.expand_as(y.tensor)] == 7.0)
    assert torch.equal(rb.tensor, x)  # out of place


# [test_complex_multi_dim_ragged_batch_input(_transposed), test_complex_multi_dim_inverse_indexing(_transposed)]
@pytest.mark.parametrize("dev", CPU_AND_GPU)
@pytest.mark.parametrize("transposed", [False, True])
def test_bool_indexing_multi_batch_dims_and_transposed_non_uniform_dim(dev, transposed):
    g = np.random.RandomState(3)
    bs, n, c = (2, 3), 9, 4
    data = g.rand(*bs, n, c).astype(np.float32)
    sizes = g.randint(0, n + 1, bs)
    mask = g.rand(*bs, n) > 0.4
    t = torch.from_numpy(data).to(dev)
    if transposed:
        d_rb = RaggedBatch(t.transpose(2, 3).contiguous(), sample_sizes=torch.from_numpy(sizes).to(dev), non_uniform_dim=3)
    else:
        d_rb = RaggedBatch(t, sample_sizes=torch.from_numpy(sizes).to(dev))
    m_rb = RaggedBatch(torch.from_numpy(mask).to(dev), sample_sizes=torch.from_numpy(sizes).to(dev))
    out = batched_bool_indexing(d_rb, m_rb)
    exp, es = oracle.bool_compact(data.reshape(6, n, c), mask.reshape(6, n), sizes.reshape(6))
    assert out.num_batch_dims == 2 and tuple(out.batch_shape) == bs
    assert out.non_uniform_dim == (3 if transposed else 2)
    got = out.get_non_uniform_dimension_transposed_to(2).tensor
    assert np.array_equal(got.cpu().numpy().reshape(exp.shape), exp)
    assert np.array_equal(out.sample_sizes.cpu().numpy().reshape(-1), es)
    # inverse: write the compacted entries back into a fresh target
    into = g.rand(*bs, n, c).astype(np.float32)
    it = torch.from_numpy(into).to(dev)
    into_rb = RaggedBatch(it.transpose(2, 3).contiguous() if transposed else it,
                          sample_sizes=torch.from_numpy(sizes).to(dev), non_uniform_dim=3 if transposed else 2)
    back = batched_bool_indexing_write(out, m_rb, into_rb)
    eb = oracle.bool_write(exp, es, mask.reshape(6, n), into.reshape(6, n, c), sizes.reshape(6))
    got_b = back.get_non_uniform_dimension_transposed_to(2).tensor
    assert isinstance(back, RaggedBatch) and back.non_uniform_dim == into_rb.non_uniform_dim
    assert np.array_equal(got_b.cpu().numpy().reshape(eb.shape), eb)


# [test_tensor_input, test_inverse_indexing_tensor_mask, test_inverse_indexing_tensor_to_write_into, test_all/empty_selections]
@pytest.mark.parametrize("dev", CPU_AND_GPU)
def test_bool_indexing_tensor_combinations_all_and_empty(dev):
    d = torch.arange(24, dtype=torch.float32).reshape(2, 4, 3).to(dev)
    all_on = torch.ones(2, 4, dtype=torch.bool, device=dev)
    none_on = torch.zeros(2, 4, dtype=torch.bool, device=dev)
    a = batched_bool_indexing(d, all_on)
    assert torch.equal(a.tensor, d) and a.sample_sizes.tolist() == [4, 4]
    e = batched_bool_indexing(d, none_on)
    assert e.shape == (2, 0, 3) and e.sample_sizes.tolist() == [0, 0]
    sizes = torch.tensor([2, 3], device=dev)
    rb = RaggedBatch(d, sample_sizes=sizes)
    r = batched_bool_indexing(rb, all_on)                 # tensor mask is limited by the data's sample sizes
    assert r.sample_sizes.tolist() == [2, 3]
    m_rb = RaggedBatch(all_on, sample_sizes=sizes)
    r2 = batched_bool_indexing(d, m_rb)                   # ragged mask limits a tensor input
    assert r2.sample_sizes.tolist() == [2, 3] and torch.equal(r2.tensor[1, :3], d[1, :3])
    w = batched_bool_indexing_write(r2, m_rb, torch.full_like(d, -1.0))
    assert isinstance(w, torch.Tensor) and torch.equal(w[0, :2], d[0, :2]) and torch.all(w[0, 2:] == -1)
    assert a.tensor.dtype == d.dtype                      # [test_dtype_consistency]
    with pytest.raises(AssertionError):
        batched_bool_indexing(d, all_on[:1])
    with pytest.raises(AssertionError):
        batched_bool_indexing_write(d, all_on, d)        # to_write must be a RaggedBatch


# [test_ragged_batch_repr, _indexing, _type_conversion, _repeat_samples(_sequence/_edge_cases), _gradients, _properties]
def test_ragged_batch_conveniences_cpu():
    t = torch.arange(24, dtype=torch.float32).reshape(2, 4, 3)
    rb = RaggedBatch(t.clone(), sample_sizes=torch.tensor([1, 4]))
    assert "RaggedBatch(" in repr(rb) and "uninitialized" in repr(rb)
    _ = rb.mask
    assert "uninitialized" not in repr(rb).split("sample_sizes")[0]
    assert torch.equal(rb[1, 2], t[1, 2])
    rb[0, 0] = torch.tensor([9.0, 9.0, 9.0])
    assert rb.tensor[0, 0].tolist() == [9.0, 9.0, 9.0]
    for name, dt in (("int", torch.int32), ("long", torch.int64), ("half", torch.float16), ("bfloat16", torch.bfloat16),
                     ("double", torch.float64), ("bool", torch.bool), ("cfloat", torch.complex64)):
        conv = getattr(rb, name)()
        assert conv.dtype == dt and torch.equal(conv.sample_sizes, rb.sample_sizes)
    rep = rb.repeat_samples([3])
    assert rep.shape == (6, 4, 3) and rep.sample_sizes.tolist() == [1, 4, 1, 4, 1, 4]
    two = RaggedBatch(t.reshape(1, 2, 4, 3), sample_sizes=torch.tensor([[1, 4]]))
    assert two.repeat_samples(2, 1).sample_sizes.tolist() == [[1, 4, 1, 4]]
    with pytest.raises(AssertionError):
        two.repeat_samples([2])                           # one factor per batch dimension
    with pytest.raises(AssertionError):
        two.repeat_samples([2, 2], 0)
    assert two.broadcast_batch_dims_to_shape((3, 2)).shape == (3, 2, 4, 3)
    with pytest.raises(AssertionError):
        two.broadcast_batch_dims_to_shape((3, 3))
    x = torch.rand(2, 4, 3, requires_grad=True)
    r = RaggedBatch(x, sample_sizes=torch.tensor([1, 4]))
    assert r.requires_grad and not r.retains_grad
    s = r.apply(lambda v: v * 2)
    s.retain_grad()
    s.tensor.sum().backward()
    assert torch.all(x.grad == 2) and s.retains_grad
    assert r.detach().requires_grad is False
    assert r.size(0) == 2 and r.dim() == 3 and r.device.type == "cpu" and r.total_num_samples_in_batch == 2
    r2 = RaggedBatch(torch.zeros(2, 4, 3), mask=torch.tensor([[1, 0, 0, 0], [1, 1, 1, 1]]).bool())
    assert r2.sample_sizes.tolist() == [1, 4]
    with pytest.raises(AssertionError):
        r.set_tensor(torch.zeros(3, 4, 3))
    with pytest.raises(ValueError):
        r.squeeze_batch_dim(0)
    with pytest.raises(ValueError):
        r.apply(lambda a, b, c, d: a)


# [test_batched_index_mapping_multi_batch_dims_forward/backward_random_runs, _multiple_batch_dims_*_manual_example]
@pytest.mark.parametrize("dev", GPU_ONLY)
def test_index_mapping_multi_batch_dims_forward_backward(dev):
    g = np.random.RandomState(8)
    for trial in range(10):
        bs = (2, 3)
        ws, wd, c = int(g.randint(2, 9)), int(g.randint(2, 9)), 3
        k = int(g.randint(1, wd + 1))
        src = g.randn(*bs, ws, c).astype(np.float32)
        into = g.randn(*bs, wd, c).astype(np.float32)
        si = g.randint(0, ws, bs + (k,))
        di = np.stack([g.permutation(wd)[:k] for _ in range(6)]).reshape(bs + (k,))
        counts = g.randint(0, k + 1, bs)
        s = torch.from_numpy(src).to(dev).requires_grad_(True)
        t = torch.from_numpy(into).to(dev).requires_grad_(True)
        out = batched_index_mapping(s, RaggedBatch(torch.from_numpy(si).to(dev), sample_sizes=torch.from_numpy(counts).to(dev)),
                                    RaggedBatch(torch.from_numpy(di).to(dev), sample_sizes=torch.from_numpy(counts).to(dev)), t)
        exp = oracle.map_pairs(src.reshape(6, ws, c), si.reshape(6, k), di.reshape(6, k), counts.reshape(6),
                               into.reshape(6, wd, c))
        assert np.array_equal(out.detach().cpu().numpy().reshape(exp.shape), exp)
        wgt = torch.from_numpy(g.randn(*out.shape).astype(np.float32)).to(dev)
        (out * wgt).sum().backward()
        gs = oracle.map_pairs(wgt.cpu().numpy().reshape(6, wd, c), di.reshape(6, k), si.reshape(6, k), counts.reshape(6),
                              np.zeros((6, ws, c), np.float32), True)
        gt = oracle.insert_const(0.0, di.reshape(6, k), counts.reshape(6), wgt.cpu().numpy().reshape(6, wd, c))
        assert np.abs(s.grad.cpu().numpy().reshape(gs.shape) - gs).max() < 1e-5
        assert np.array_equal(t.grad.cpu().numpy().reshape(gt.shape), gt)


# [test_indexing_write_backward_*, test_inverse_indexing_backward_*, test_gradient_propagation]
@pytest.mark.parametrize("dev", GPU_ONLY)
def test_gradients_through_scatter_ops(dev):
    g = np.random.RandomState(4)
    b, w, k, c = 3, 7, 4, 2
    idx = np.stack([g.permutation(w)[:k] for _ in range(b)])
    counts = np.array([4, 2, 0])
    rb = RaggedBatch(torch.from_numpy(idx).to(dev), sample_sizes=torch.from_numpy(counts).to(dev))
    src = torch.from_numpy(g.randn(b, k, c).astype(np.float32)).to(dev).requires_grad_(True)
    into = torch.from_numpy(g.randn(b, w, c).astype(np.float32)).to(dev).requires_grad_(True)
    wgt = torch.from_numpy(g.randn(b, w, c).astype(np.float32)).to(dev)
    out = batched_indexing_write(src, rb, into)
    (out * wgt).sum().backward()
    exp_src = oracle.gather(wgt.cpu().numpy(), idx, counts, 0.0)          # gradient of the written entries
    exp_into = oracle.insert_const(0.0, idx, counts, wgt.cpu().numpy())    # overwritten slots get no gradient
    assert np.array_equal(src.grad.cpu().numpy(), exp_src) and np.array_equal(into.grad.cpu().numpy(), exp_into)
    src2 = src.detach().clone().requires_grad_(True)
    inv = batched_inverse_indexing_access(src2, rb, w, -3.0)
    (inv * wgt).sum().backward()
    assert np.array_equal(src2.grad.cpu().numpy(), exp_src)
    assert np.array_equal(inv.detach().cpu().numpy(), oracle.scatter_new(src.detach().cpu().numpy(), idx, counts, w, -3.0, False))
    # RaggedBatch in, RaggedBatch out
    into_rb = RaggedBatch(into.detach(), sample_sizes=torch.tensor([7, 5, 2], device=dev))
    res = batched_indexing_write(RaggedBatch(src.detach(), sample_sizes=rb.sample_sizes), rb, into_rb)
    assert isinstance(res, RaggedBatch) and torch.equal(res.sample_sizes, into_rb.sample_sizes)


# [test_batched_index_mapping_mismatched_*, _indices_wrong_num_dims, test_indexing_write_inconsistent_dims]
@pytest.mark.parametrize("dev", GPU_ONLY)
def test_error_paths_of_the_index_ops(dev):
    d = torch.zeros(2, 5, 3, device=dev)
    i64 = RaggedBatch(torch.zeros(2, 2, dtype=torch.int64, device=dev), sample_sizes=torch.ones(2, dtype=torch.int64, device=dev))
    i32 = RaggedBatch(torch.zeros(2, 2, dtype=torch.int32, device=dev), sample_sizes=torch.ones(2, dtype=torch.int32, device=dev))
    with pytest.raises(RuntimeError):
        batched_index_mapping(d, i64, i32, d)                       # index dtypes differ
    with pytest.raises(RuntimeError):
        batched_index_mapping(d.double(), i64, i64, d)              # data dtypes differ
    with pytest.raises(RuntimeError):
        batched_index_mapping(d.cpu(), i64, i64, d)                 # devices differ
    with pytest.raises(RuntimeError):
        batched_index_mapping(d[:, :, :2].contiguous(), i64, i64, d)  # trailing data sizes differ
    i3 = RaggedBatch(torch.zeros(2, 3, dtype=torch.int64, device=dev), sample_sizes=torch.ones(2, dtype=torch.int64, device=dev))
    with pytest.raises(AssertionError):
        batched_index_mapping(d, i64, i3, d)                        # index widths differ
    with pytest.raises(AssertionError):
        batched_indexing_access(d, i64, dim_to_index_in=0)          # cannot index a batch dimension
    with pytest.raises(RuntimeError):
        batched_indexing_write(torch.zeros(2, 2, 4, device=dev), i64, d)   # inconsistent trailing dims
    m = get_mask_from_indices(5, i64)
    assert m.shape == (2, 5) and m[:, 0].all() and not m[:, 1:].any()


@pytest.mark.parametrize("dev", GPU_ONLY)
def test_no_grad_results_do_not_require_grad(dev):
    # under torch.no_grad() Function.apply returns tensors that do not require grad even when an input
    # does (a parameter); the no-autograd fast path has to behave the same
    x = torch.rand(2, 5, 3, device=dev, requires_grad=True)
    idx = RaggedBatch(torch.tensor([[0, 2, 4], [1, 3, 0]], device=dev), sample_sizes=torch.tensor([3, 2], device=dev))
    into = torch.zeros(2, 6, 3, device=dev, requires_grad=True)
    with torch.no_grad():
        a = batched_indexing_access(x, idx, 0.5)
        b = batched_inverse_indexing_access(x[:, :3], idx, 6, 0.0)
        c = batched_indexing_write(x[:, :3], idx, into)
        d = batched_index_mapping(x, idx, idx, into)
    for t in (a.tensor, b, c, d):
        assert t.requires_grad is False and t.grad_fn is None
        t.add_(1.0)                                   # in-place edits of such results are legal
    # with grad mode on the same calls are differentiable
    a = batched_indexing_access(x, idx, 0.5)
    assert a.tensor.requires_grad and a.tensor.grad_fn is not None
