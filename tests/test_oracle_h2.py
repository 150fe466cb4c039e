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
