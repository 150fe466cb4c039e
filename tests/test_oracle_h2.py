"""Pins oracle/h2.py against the reference's hand-written expected tensors (tests/h2_literals.py) and checks the
CPU product paths of accvlab.batching_helpers (RaggedBatch, combine_data/split = config C0, compaction,
reductions) against the oracle.  CPU only."""
import numpy as np
import pytest
import torch

import bench_workloads as wl
import h2_literals as lit
from oracle import h2 as oracle
from accvlab.batching_helpers import (RaggedBatch, apply_mask_to_tensor, average_over_targets, batched_bool_indexing,
                                      batched_bool_indexing_write, combine_data, get_compact_from_named_tuple,
                                      get_compact_lists, get_indices_from_mask, squeeze_except_batch_and_sample,
                                      sum_over_targets)


def test_oracle_gather_and_grad_literal():
    data, idx, counts, fill, exp, grad = lit.gather_literal()
    out = oracle.gather(data.reshape(6, 5, 2), idx.reshape(6, 3), counts.reshape(6), fill).reshape(2, 3, 3, 2)
    assert np.array_equal(out, exp)
    # backward of sum(sin(gathered)) == accumulate-scatter of cos(gathered)
    g = oracle.scatter_new(np.cos(out).reshape(6, 3, 2), idx.reshape(6, 3), counts.reshape(6), 5, 0.0, True)
    assert np.allclose(g.reshape(2, 3, 5, 2), grad, atol=1e-6)


def test_oracle_mask_from_indices_literal():
    idx, counts, n, exp = lit.mask_from_indices_literal()
    assert np.array_equal(oracle.mask_from_indices(idx, counts, n), exp)


def test_oracle_compaction_literal():
    mask, datas, sizes, exps = lit.compaction_literal()
    for d, e in zip(datas, exps):
        out, s = oracle.bool_compact(d, mask)
        assert np.array_equal(s, sizes) and np.array_equal(out, e)


def test_oracle_mapping_literal():
    src, si, di, counts, into, exp = lit.mapping_literal()
    assert np.array_equal(oracle.map_pairs(src, si, di, counts, into), exp)


def test_oracle_accumulate_first_write_replaces():
    src = np.array([[[1.0], [2.0], [4.0]]])
    idx = np.array([[1, 1, 1]])
    out = oracle.scatter_new(src, idx, np.array([3]), 3, -7.0, True)
    assert out[0, :, 0].tolist() == [-7.0, 7.0, -7.0]
    out = oracle.scatter_new(src, idx, np.array([3]), 3, -7.0, False)
    assert out[0, 1, 0] in (1.0, 2.0, 4.0)


# ----------------------------------------------------------------------------- C0: pack / mask / unpack on CPU
def test_c0_pack_mask_split_cpu():
    boxes = wl.ragged_boxes(64, 1, 32, seed=0)
    rb = combine_data(boxes)
    exp, sizes = oracle.combine([b.numpy() for b in boxes])
    assert rb.tensor.dtype == torch.float32 and rb.tensor.shape == exp.shape
    assert np.array_equal(rb.tensor.numpy(), exp)                      # bit exact
    assert rb.sample_sizes.dtype == torch.int64 and np.array_equal(rb.sample_sizes.numpy(), sizes)
    m = rb.mask
    assert m.dtype == torch.bool and m.shape == (64, exp.shape[1])
    assert np.array_equal(m.numpy(), np.arange(exp.shape[1])[None, :] < sizes[:, None])
    parts = rb.split()
    assert len(parts) == 64 and all(torch.equal(a, b) for a, b in zip(parts, boxes))
    assert rb.total_num_entries == int(sizes.sum()) and rb.max_sample_size == exp.shape[1]


def test_combine_nested_and_shared_sizes_and_empty():
    boxes = wl.ragged_boxes(6, 0, 5, seed=3)
    nested = [[boxes[0], boxes[1], boxes[2]], [boxes[3], boxes[4], boxes[5]]]
    flat = combine_data(nested)
    assert flat.num_batch_dims == 1 and flat.shape[0] == 6
    nb = combine_data(nested, flatten_batch_dims=False)
    assert nb.num_batch_dims == 2 and tuple(nb.batch_shape) == (2, 3) and nb.non_uniform_dim == 2
    exp, sizes = oracle.combine([b.numpy() for b in boxes])
    assert np.array_equal(nb.tensor.numpy().reshape(exp.shape), exp)
    assert np.array_equal(nb.sample_sizes.numpy().reshape(-1), sizes)
    got = nb.split()
    assert torch.equal(got[1][2], boxes[5]) and len(got) == 2 and len(got[0]) == 3
    labels = [torch.arange(b.shape[0]) for b in boxes]
    lab = combine_data(labels, other_with_same_sample_sizes=flat)
    assert lab.mask is flat.mask or torch.equal(lab.mask, flat.mask)
    assert lab.tensor.dtype == torch.int64
    with pytest.raises(AssertionError):
        combine_data([boxes[0], "x"])
    with pytest.raises(AssertionError):
        combine_data([[boxes[0]], [boxes[1], boxes[2]]], flatten_batch_dims=False)
    with pytest.raises(AssertionError):
        combine_data([])
    e = combine_data([torch.zeros(0, 4), torch.zeros(0, 4)])
    assert e.shape == (2, 0, 4) and e.sample_sizes.tolist() == [0, 0]


def test_combine_gradients_flow_on_cpu():
    a = torch.rand(3, 2, requires_grad=True)
    b = torch.rand(1, 2, requires_grad=True)
    rb = combine_data([a, b])
    rb.tensor.sum().backward()
    assert torch.all(a.grad == 1) and torch.all(b.grad == 1)


def test_bool_indexing_cpu_matches_oracle_and_literals():
    mask, datas, sizes, exps = lit.compaction_literal()
    out = get_compact_lists(torch.from_numpy(mask), [torch.from_numpy(datas[0]), "keep", torch.from_numpy(datas[1])])
    assert out[1] == "keep"
    for rb, e in zip((out[0], out[2]), exps):
        assert isinstance(rb, RaggedBatch)
        assert np.array_equal(rb.tensor.numpy(), e) and np.array_equal(rb.sample_sizes.numpy(), sizes)
    idx = get_indices_from_mask(torch.from_numpy(mask))
    eidx, esz = oracle.indices_from_mask(mask)
    assert np.array_equal(idx.tensor.numpy(), eidx) and np.array_equal(idx.sample_sizes.numpy(), esz)
    g = np.random.RandomState(0)
    for trial in range(20):
        b, m, k = g.randint(1, 6), g.randint(1, 12), g.randint(1, 4)
        d = g.rand(b, m, k).astype(np.float32)
        mk = g.rand(b, m) > 0.5
        valid = g.randint(0, m + 1, b)
        rb_in = RaggedBatch(torch.from_numpy(d), sample_sizes=torch.from_numpy(valid))
        got = batched_bool_indexing(rb_in, torch.from_numpy(mk))
        exp, es = oracle.bool_compact(d, mk, valid)
        assert np.array_equal(got.tensor.numpy(), exp) and np.array_equal(got.sample_sizes.numpy(), es)
        into = g.rand(b, m, k).astype(np.float32)
        back = batched_bool_indexing_write(got, torch.from_numpy(mk), RaggedBatch(torch.from_numpy(into),
                                                                                  sample_sizes=torch.from_numpy(valid)))
        assert np.array_equal(back.tensor.numpy(), oracle.bool_write(exp, es, mk, into, valid))


def test_ragged_batch_shape_ops_and_reductions_cpu():
    boxes = wl.ragged_boxes(6, 1, 5, seed=1)
    rb = combine_data(boxes)
    sizes = rb.sample_sizes
    t = rb.get_non_uniform_dimension_transposed_to(2)
    assert t.non_uniform_dim == 2 and t.shape == (6, 4, rb.max_sample_size)
    assert all(torch.equal(a, b.t()) for a, b in zip(t.split(), boxes))  # samples keep the tensor's layout
    w = t.get_existence_weights()
    assert w.shape == t.shape and torch.equal(w[:, 0, :].bool(), rb.mask)
    filled = rb.with_padded_set_to(-1.0)
    assert np.array_equal(filled.tensor.numpy(), oracle.pad_fill(rb.tensor.numpy(), sizes.numpy(), -1.0))
    assert torch.equal(rb.tensor, combine_data(boxes).tensor)  # out of place
    s = sum_over_targets(filled)
    assert torch.allclose(s, torch.stack([b.sum(0) for b in boxes]), atol=1e-6)
    a = average_over_targets(filled)
    assert torch.allclose(a, torch.stack([b.mean(0) for b in boxes]), atol=1e-6)
    r = rb.reshape_batch_dims((2, 3))
    assert r.num_batch_dims == 2 and r.non_uniform_dim == 2 and r.flatten_batch_dims().shape == rb.shape
    u = rb.unsqueeze_batch_dim(0)
    assert u.shape[0] == 1 and u.squeeze_batch_dim(0).shape == rb.shape
    assert rb.repeat_samples(2, 0).shape[0] == 12
    assert rb.unsqueeze_data_dim(1).non_uniform_dim == 2
    assert rb.to(torch.float64).dtype == torch.float64 and rb.long().dtype == torch.int64
    doubled = rb.apply(lambda x: x * 2)
    assert torch.equal(doubled.tensor, rb.tensor * 2)
    sq = squeeze_except_batch_and_sample(RaggedBatch(rb.tensor.reshape(6, 1, -1, 4, 1), sample_sizes=sizes,
                                                     non_uniform_dim=2))
    assert sq.shape == rb.shape and sq.non_uniform_dim == 1
    m = apply_mask_to_tensor(rb.tensor, rb.mask, 9.0)
    assert np.array_equal(m.numpy(), oracle.pad_fill(rb.tensor.numpy(), sizes.numpy(), 9.0))
    full = RaggedBatch.FromFullTensor(torch.zeros(3, 4, 2))
    assert full.sample_sizes.tolist() == [4, 4, 4] and bool(full.mask.all())
    over = RaggedBatch.FromOversizeTensor(torch.zeros(2, 9, 3), sample_sizes=torch.tensor([2, 5]))
    assert over.shape == (2, 5, 3)
    e = RaggedBatch.Empty(3, 1, "cpu")
    assert e.shape == (0, 0, 0)
    with pytest.raises(AssertionError):
        RaggedBatch(torch.zeros(2, 3))
    with pytest.raises(AssertionError):
        RaggedBatch(torch.zeros(2, 3), sample_sizes=torch.zeros(3, dtype=torch.int64))
    b1, b2 = RaggedBatch.broadcast_batch_dims([rb, combine_data(boxes[:1])])
    assert b1.shape[0] == b2.shape[0] == 6


def test_named_tuple_compaction():
    from collections import namedtuple
    T = namedtuple("T", ["a", "n", "b"])
    mask, datas, sizes, exps = lit.compaction_literal()
    out = get_compact_from_named_tuple(torch.from_numpy(mask), T(torch.from_numpy(datas[0]), 3, torch.from_numpy(datas[1])))
    assert isinstance(out, T) and out.n == 3 and np.array_equal(out.b.tensor.numpy(), exps[1])


# ----------------------------------------------------------------------------- round 2: the remaining reference literals
@pytest.mark.parametrize("fill", [0.0, 1.0, 2.3])
def test_oracle_inverse_literal(fill):
    data, idx, counts, n_targets, exp, grad = lit.inverse_literal(fill)
    out = oracle.scatter_new(data.reshape(6, 4), idx.reshape(6, 4), counts.reshape(6), n_targets, fill, False)
    assert np.array_equal(out.reshape(2, 3, 5), exp)
    # backward = gather of the incoming gradient with 0 filler (batched_indexing_ops.py:72-117)
    g = oracle.gather(np.cos(exp.astype(np.float64)).reshape(6, 5), idx.reshape(6, 4), counts.reshape(6), 0.0)
    assert np.allclose(g.reshape(2, 3, 4), grad, atol=1e-6)


def test_oracle_write_literal():
    data, idx, counts, into, exp, g_src, g_into = lit.write_literal()
    out = oracle.scatter_insert(data.reshape(6, 4), idx.reshape(6, 4), counts.reshape(6), into.reshape(6, 5))
    assert np.array_equal(out.reshape(2, 3, 5), exp)
    up = np.cos(exp.astype(np.float64)).reshape(6, 5)
    assert np.allclose(oracle.gather(up, idx.reshape(6, 4), counts.reshape(6), 0.0).reshape(2, 3, 4), g_src, atol=1e-6)
    assert np.allclose(oracle.insert_const(0.0, idx.reshape(6, 4), counts.reshape(6), up).reshape(2, 3, 5), g_into, atol=1e-6)


@pytest.mark.parametrize("multi", [False, True])
def test_oracle_bool_index_and_write_literals(multi):
    data, sizes, mask, exp, out_sizes = (lit.bool_index_multi_literal if multi else lit.bool_index_simple_literal)()
    b = int(np.prod(sizes.shape))
    flat = lambda a, w: a.reshape((b, w) + a.shape[sizes.ndim + 1:])  # noqa: E731
    out, s = oracle.bool_compact(flat(data, data.shape[sizes.ndim]), mask.reshape(b, -1), sizes.reshape(b))
    assert np.array_equal(s, out_sizes.reshape(b)) and np.array_equal(out, flat(exp, exp.shape[sizes.ndim]))
    compact, csz, mask, sizes, into, want = lit.bool_write_literal(multi)
    back = oracle.bool_write(flat(compact, compact.shape[sizes.ndim]), csz.reshape(b), mask.reshape(b, -1),
                             flat(into, into.shape[sizes.ndim]), sizes.reshape(b))
    assert np.array_equal(back, flat(want, want.shape[sizes.ndim]))


@pytest.mark.parametrize("multi", [False, True])
def test_oracle_pad_fill_literal(multi):
    data, sizes, value, exp, grad = lit.pad_fill_literal(multi)
    out = oracle.pad_fill(data.reshape(-1, 5), sizes.reshape(-1), value)
    assert np.array_equal(out.reshape(exp.shape), exp)
    assert np.allclose(oracle.pad_fill(np.cos(data.astype(np.float64)).reshape(-1, 5), sizes.reshape(-1), 0.0)
                       .reshape(exp.shape), grad, atol=1e-6)


def test_oracle_indices_from_mask_literals():
    for mask, sizes, rows, width in lit.indices_from_mask_literals():
        out, s = oracle.indices_from_mask(mask, sizes)
        assert s.tolist() == [len(r) for r in rows]
        assert out.shape[1] == max(len(r) for r in rows)          # the oracle's width is the max count
        for i, r in enumerate(rows):
            assert out[i, :len(r)].tolist() == r


def test_oracle_mapping_grads_and_multi_batch_literal():
    src, si, di, counts, into, exp = lit.mapping_literal()
    g_src, g_into = lit.mapping_grads_literal()
    up = np.cos(exp.astype(np.float64))
    # d(src): accumulate-scatter of the upstream gradient read at the destination slots (batched_index_mapping_op.py:54-81)
    got_src = oracle.map_pairs(up, di, si, counts, np.zeros_like(src, dtype=np.float64), accumulate=True)
    assert np.allclose(got_src, g_src, atol=1e-6)
    assert np.allclose(oracle.insert_const(0.0, di, counts, up), g_into, atol=1e-6)
    src, si, di, counts, into, exp, g_src, g_into = lit.mapping_multi_batch_literal()
    out = oracle.map_pairs(src.reshape(4, 3, 2), si.reshape(4, 3), di.reshape(4, 3), counts.reshape(4), into.reshape(4, 4, 2))
    assert np.array_equal(out.reshape(exp.shape), exp)


def test_cpu_product_paths_on_the_new_literals():
    # the CPU paths the reference itself has (RaggedBatch pad fill, boolean compaction / write-back, indices from mask,
    # combine_data shapes) against the same literals
    for multi in (False, True):
        data, sizes, value, exp, _ = lit.pad_fill_literal(multi)
        rb = RaggedBatch(torch.from_numpy(data.copy()), sample_sizes=torch.from_numpy(sizes))
        assert np.array_equal(rb.with_padded_set_to(value).tensor.numpy(), exp)
        data, sizes, mask, exp, out_sizes = (lit.bool_index_multi_literal if multi else lit.bool_index_simple_literal)()
        rb = RaggedBatch(torch.from_numpy(data), sample_sizes=torch.from_numpy(sizes))
        mrb = RaggedBatch(torch.from_numpy(mask), sample_sizes=torch.from_numpy(sizes))
        got = batched_bool_indexing(rb, mrb)
        assert np.array_equal(got.tensor.numpy(), exp) and np.array_equal(got.sample_sizes.numpy(), out_sizes)
        compact, csz, mask, sizes, into, want = lit.bool_write_literal(multi)
        back = batched_bool_indexing_write(RaggedBatch(torch.from_numpy(compact), sample_sizes=torch.from_numpy(csz)),
                                           RaggedBatch(torch.from_numpy(mask), sample_sizes=torch.from_numpy(sizes)),
                                           RaggedBatch(torch.from_numpy(into), sample_sizes=torch.from_numpy(sizes)))
        assert np.array_equal(back.tensor.numpy(), want)
    (m1, _, rows1, w1), (m2, s2, rows2, w2) = lit.indices_from_mask_literals()
    i1 = get_indices_from_mask(torch.from_numpy(m1))
    assert i1.tensor.dtype == torch.int64 and tuple(i1.tensor.shape) == (4, w1)
    i2 = get_indices_from_mask(RaggedBatch(torch.from_numpy(m2), sample_sizes=torch.from_numpy(s2)))
    assert tuple(i2.tensor.shape) == (3, w2)
    for got, rows in ((i1, rows1), (i2, rows2)):
        assert got.sample_sizes.tolist() == [len(r) for r in rows]
        for i, r in enumerate(rows):
            assert got.tensor[i, :len(r)].tolist() == r
    sh = lit.combine_literal_shapes()
    g = torch.Generator().manual_seed(0)
    flat = [torch.randn(n, generator=g) for n in sh["flat"]]
    c = combine_data(flat)
    assert tuple(c.tensor.shape) == (3, 4) and all(torch.equal(c.tensor[i, :n], flat[i]) for i, n in enumerate(sh["flat"]))
    cn = combine_data([[flat[0], flat[1]], [flat[2]]])
    assert torch.equal(cn.tensor, c.tensor)
    extra = [torch.randn(s, generator=g) for s in sh["extra"]]
    ce = combine_data(extra)
    assert tuple(ce.tensor.shape) == (2, 3, 4) and torch.equal(ce.tensor[0, :2], extra[0]) and torch.equal(ce.tensor[1], extra[1])
    grid = [[torch.randn((n,) + sh["grid_inner"], generator=g) for n in row] for row in sh["grid"]]
    cg = combine_data(grid, flatten_batch_dims=False)
    assert tuple(cg.batch_shape) == (2, 3) and tuple(cg.shape) == (2, 3, 7, 3, 4)
    for i, row in enumerate(sh["grid"]):
        for j, n in enumerate(row):
            assert int(cg.sample_sizes[i, j]) == n and torch.equal(cg.tensor[i, j, :n], grid[i][j])
    with pytest.raises(AssertionError):
        combine_data([grid[0], grid[1] + [torch.randn(3, 3, 4)]], flatten_batch_dims=False)
