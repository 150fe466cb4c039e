"""Seeded random sweeps of the CPU side of batching_helpers (config C0 runs here: torch-CPU tensors, the host extension and its
python twin) against the oracle: pack / mask / split round trips over random sample lengths, inner shapes and dtypes incl. empty
samples and nested batch dimensions; compaction and pad fill of CPU tensors."""
import numpy as np
import pytest


def _seeds(n):
    """seed range of a sweep; ACCV_FUZZ_SCALE=k runs k times as many seeds (soak runs: profiles/r03_fuzz_soak.log)"""
    import os
    return range(n * max(1, int(os.environ.get("ACCV_FUZZ_SCALE", "1"))))

import torch

from oracle import h2 as oracle_h2

_DTYPES = [torch.float32, torch.float64, torch.int64, torch.int32, torch.float16, torch.uint8, torch.bool]


def _rand(shape, dt, rng):
    a = rng.random(shape) * 60
    return torch.from_numpy(a > 30) if dt == torch.bool else torch.from_numpy(a).to(dt)


@pytest.mark.parametrize("seed", _seeds(10))
def test_combine_mask_split_round_trip_random(seed):
    from accvlab.batching_helpers import combine_data

    rng = np.random.default_rng(8000 + seed)
    for case in range(8):
        b = int(rng.integers(1, 9))
        inner = tuple(int(x) for x in rng.integers(1, 5, size=int(rng.integers(0, 3))))
        dt = _DTYPES[(seed + case) % len(_DTYPES)]
        lens = rng.integers(0, 7, size=b)
        samples = [_rand((int(n),) + inner, dt, rng) for n in lens]
        rb = combine_data(samples)
        want, sizes = oracle_h2.combine([s.numpy() for s in samples])
        assert np.array_equal(rb.sample_sizes.numpy(), sizes)
        assert rb.tensor.dtype == dt and np.array_equal(rb.tensor.numpy(), want)
        cols = np.arange(want.shape[1])[None, :]
        assert np.array_equal(rb.mask.numpy(), cols < sizes[:, None])
        back = rb.split()
        assert len(back) == b and all(torch.equal(x, y) for x, y in zip(back, samples))
        # pad fill on CPU tensors (batched_indexing_access_cpu_impl.cpp:27-44)
        if dt != torch.bool:
            filled = rb.with_padded_set_to(5)
            assert np.array_equal(filled.tensor.numpy(), oracle_h2.pad_fill(want, sizes, 5))


@pytest.mark.parametrize("seed", _seeds(6))
def test_nested_batch_dims_round_trip_random(seed):
    from accvlab.batching_helpers import combine_data

    rng = np.random.default_rng(9000 + seed)
    for case in range(5):
        outer, inner_b = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        feat = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(0, 2))))
        dt = _DTYPES[(seed + case) % 4]
        nested = [[_rand((int(rng.integers(0, 5)),) + feat, dt, rng) for _ in range(inner_b)] for _ in range(outer)]
        rb = combine_data(nested, flatten_batch_dims=False)
        assert rb.num_batch_dims == 2 and tuple(rb.sample_sizes.shape) == (outer, inner_b)
        if all(t.shape[0] == 0 for row in nested for t in row):
            # nothing but empty samples: the reference returns an empty float32 batch without the trailing dimensions
            # (batched_processing_py.py:524-533) — found by the soak run (ACCV_FUZZ_SCALE), kept as the reference has it
            assert tuple(rb.tensor.shape) == (outer, inner_b, 0) and rb.tensor.dtype == torch.float32
            assert int(rb.sample_sizes.sum()) == 0
            continue
        for i in range(outer):
            for j in range(inner_b):
                n = nested[i][j].shape[0]
                assert int(rb.sample_sizes[i, j]) == n
                assert torch.equal(rb.tensor[i, j, :n], nested[i][j])
                assert not bool(rb.tensor[i, j, n:].to(torch.float64).abs().sum())        # zero padding
        back = rb.split()
        assert all(torch.equal(back[i][j], nested[i][j]) for i in range(outer) for j in range(inner_b))
        flat = combine_data(nested)          # default: nesting flattened depth-first into one batch dimension
        assert flat.num_batch_dims == 1 and flat.sample_sizes.tolist() == [t.shape[0] for row in nested for t in row]


@pytest.mark.parametrize("seed", _seeds(6))
def test_cpu_compaction_random_against_the_oracle(seed):
    from accvlab.batching_helpers import batched_bool_indexing, get_compact_lists, get_indices_from_mask

    rng = np.random.default_rng(9500 + seed)
    for case in range(6):
        b, m = int(rng.integers(1, 7)), int(rng.choice([1, 5, 64, 130]))
        inner = tuple(int(x) for x in rng.integers(1, 4, size=int(rng.integers(0, 2))))
        dt = _DTYPES[(seed + case) % len(_DTYPES)]
        data = _rand((b, m) + inner, dt, rng)
        mask = torch.from_numpy(rng.random((b, m)) < rng.random())
        want, sizes = oracle_h2.bool_compact(data.numpy(), mask.numpy())
        for got in (batched_bool_indexing(data, mask), get_compact_lists(mask, [data])[0]):
            assert np.array_equal(got.sample_sizes.numpy(), sizes)
            assert np.array_equal(got.tensor.numpy()[:, : want.shape[1]], want)
        idx = get_indices_from_mask(mask)
        want_idx, want_cnt = oracle_h2.indices_from_mask(mask.numpy())
        assert np.array_equal(idx.sample_sizes.numpy(), want_cnt) and np.array_equal(idx.tensor.numpy()[:, : want_idx.shape[1]], want_idx)


# ---------------------------------------------------------------------------------------------------------------------
# model-based sweep of RaggedBatch's shape operations: the model of a batch is the nested list of its samples' valid
# entries (what split() returns); every operation has an obvious meaning on that list
def _leaves(nested, depth):
    return [nested] if depth == 0 else [l for sub in nested for l in _leaves(sub, depth - 1)]


def _random_ragged(rng, nb=None):
    from accvlab.batching_helpers import RaggedBatch

    nb = int(rng.integers(1, 4)) if nb is None else nb
    batch_shape = tuple(int(x) for x in rng.integers(1, 4, size=nb))
    n_data = int(rng.integers(1, 4))                      # data dims incl. the non-uniform one
    nu_pos = int(rng.integers(0, n_data))
    data_shape = [int(x) for x in rng.integers(1, 4, size=n_data)]
    sizes = rng.integers(0, 5, size=batch_shape)
    longest = int(sizes.max())
    data_shape[nu_pos] = longest
    t = torch.from_numpy(rng.random(batch_shape + tuple(data_shape)))
    nu = nb + nu_pos
    # junk in the padding must never be visible through split()
    return RaggedBatch(t, sample_sizes=torch.from_numpy(sizes), non_uniform_dim=nu), nb, nu_pos


def _model(rbatch):
    nb, nu = rbatch.num_batch_dims, rbatch.non_uniform_dim
    out = []
    flat_t = rbatch.tensor.reshape((-1,) + tuple(rbatch.tensor.shape[nb:])) if rbatch.tensor.numel() else None
    sizes = rbatch.sample_sizes.reshape(-1).tolist()
    for i, n in enumerate(sizes):
        if flat_t is None:
            shape = list(rbatch.tensor.shape[nb:])
            shape[nu - nb] = 0
            out.append(torch.zeros(shape, dtype=rbatch.tensor.dtype))
        else:
            out.append(flat_t[i].narrow(nu - nb, 0, n))
    return out


def _check(rbatch, want_leaves, want_batch_shape, what):
    assert tuple(rbatch.batch_shape) == tuple(want_batch_shape), what
    assert tuple(rbatch.sample_sizes.shape) == tuple(want_batch_shape), what
    got = _leaves(rbatch.split(), rbatch.num_batch_dims)
    assert len(got) == len(want_leaves), what
    for g, w in zip(got, want_leaves):
        assert tuple(g.shape) == tuple(w.shape) and torch.equal(g, w), what
    cols = torch.arange(rbatch.tensor.shape[rbatch.non_uniform_dim])
    assert torch.equal(rbatch.mask, cols < rbatch.sample_sizes.unsqueeze(-1)), what + ": mask"


@pytest.mark.parametrize("seed", _seeds(12))
def test_ragged_batch_shape_operations_against_the_list_model(seed):
    rng = np.random.default_rng(9800 + seed)
    for case in range(8):
        rbatch, nb, nu_pos = _random_ragged(rng)
        leaves = _model(rbatch)
        shape = tuple(rbatch.batch_shape)
        _check(rbatch, leaves, shape, "identity")
        # flatten / reshape of the batch dimensions keep the samples in row-major order
        _check(rbatch.flatten_batch_dims(), leaves, (len(leaves),), "flatten_batch_dims")
        total = len(leaves)
        for cand in ((total,), (1, total), (total, 1), shape[::-1]):
            if int(np.prod(cand)) == total:
                _check(rbatch.reshape_batch_dims(cand), leaves, cand, f"reshape_batch_dims{cand}")
        # unsqueeze / squeeze a batch dimension
        d = int(rng.integers(0, nb + 1))
        un = rbatch.unsqueeze_batch_dim(d)
        _check(un, leaves, shape[:d] + (1,) + shape[d:], "unsqueeze_batch_dim")
        _check(un.squeeze_batch_dim(d), leaves, shape, "squeeze_batch_dim")
        # move the non-uniform dimension: every sample is transposed accordingly
        n_data = rbatch.tensor.dim() - nb
        new_pos = int(rng.integers(0, n_data))
        moved = rbatch.get_non_uniform_dimension_transposed_to(nb + new_pos)
        assert moved.non_uniform_dim == nb + new_pos
        _check(moved, [l.transpose(nu_pos, new_pos) for l in leaves], shape, "get_non_uniform_dimension_transposed_to")
        # a new data dimension
        dd = int(rng.integers(nb, rbatch.tensor.dim() + 1))
        _check(rbatch.unsqueeze_data_dim(dd), [l.unsqueeze(dd - nb) for l in leaves], shape, "unsqueeze_data_dim")
        # repeat the samples along one batch dimension
        bd, k = int(rng.integers(0, nb)), int(rng.integers(1, 4))
        rep = rbatch.repeat_samples(k, bd)
        idx = np.arange(total).reshape(shape)
        idx = np.tile(idx, [k if a == bd else 1 for a in range(nb)])
        _check(rep, [leaves[j] for j in idx.reshape(-1)], idx.shape, "repeat_samples")
        # padding never leaks: fill it and compare again
        _check(rbatch.with_padded_set_to(-7.0), leaves, shape, "with_padded_set_to")
        # existence weights: shaped like the data, 1 for valid entries and 0 in the padding (ragged_batch.py:493-522)
        w = rbatch.get_existence_weights()
        t = rbatch.tensor
        pos = torch.arange(t.shape[rbatch.non_uniform_dim]).reshape([-1 if a == rbatch.non_uniform_dim else 1 for a in range(t.dim())])
        valid = pos < rbatch.sample_sizes.reshape(tuple(shape) + (1,) * (t.dim() - nb))
        assert w.shape == t.shape and torch.equal(w, valid.expand(t.shape).to(w.dtype))


# ---------------------------------------------------------------------------------------------------------------------
def _random_tree(rng, depth=0):
    kinds = ["tensor"] * 5 + ["list", "dict", "tuple", "other", "numpy"] if depth < 3 else ["tensor", "other"]
    kind = kinds[int(rng.integers(0, len(kinds)))]
    if kind in ("tensor", "numpy"):
        dt = [torch.float32, torch.int64, torch.float16, torch.uint8, torch.bool, torch.complex64, torch.float64][int(rng.integers(0, 7))]
        shape = tuple(int(x) for x in rng.integers(0, 9, size=int(rng.integers(0, 4))))
        t = (torch.rand(shape) * 50).to(dt) if dt not in (torch.bool, torch.complex64) else \
            (torch.rand(shape) > 0.5 if dt == torch.bool else torch.complex(torch.rand(shape), torch.rand(shape)))
        if shape and rng.integers(0, 6) == 0 and t.dim() >= 2:
            t = t.transpose(0, -1)
        return t.numpy() if kind == "numpy" and dt != torch.float16 else t
    if kind == "other":
        return [None, 3, "text", 2.5][int(rng.integers(0, 4))]
    items = [_random_tree(rng, depth + 1) for _ in range(int(rng.integers(0, 5)))]
    return items if kind == "list" else tuple(items) if kind == "tuple" else {f"k{i}": v for i, v in enumerate(items)}


def _same_tree(a, b):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(a.copy())          # (np.ascontiguousarray would turn 0-d into 1-d)
    if isinstance(a, torch.Tensor):
        return isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b.cpu())
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(_same_tree(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(_same_tree(a[k], b[k]) for k in a)
    return a is b or a == b


@pytest.mark.parametrize("seed", _seeds(8))
def test_pack_batch_random_trees_round_trip(seed):
    """DataLoader hook: pack_batch -> (pickle) -> unpack reproduces nesting, dtypes, shapes and bytes for random trees with empty,
    non-contiguous, numpy and non-tensor leaves; every packed leaf starts at a multiple of the requested alignment"""
    import pickle

    from accvlab.multi_tensor_copier import pack_batch

    rng = np.random.default_rng(9990 + seed)
    for case in range(6):
        tree = [_random_tree(rng) for _ in range(int(rng.integers(1, 10)))]
        align = int(rng.choice([1, 6, 16, 64]))
        pk = pack_batch(tree, min_packed_alignment_bytes=align)
        assert _same_tree(tree, pk.unpack()), f"seed {seed} case {case}"
        again = pickle.loads(pickle.dumps(pk))
        assert _same_tree(tree, again.unpack()), f"seed {seed} case {case}: pickled"
        assert again.num_tensors == pk.num_tensors and again.num_packed == pk.num_packed


@pytest.mark.parametrize("seed", _seeds(6))
def test_lane_sampler_host_path_random_against_the_oracle(seed):
    """interpolate / lengths on CPU tensors (accv_polyline_sample_host, double accumulation as polyline_cpu.cpp:111-132):
    float32 and float64, fixed and ragged, zero-length segments, empty polylines, relative mode"""
    from oracle import lane as oracle_lane

    from accvlab.batching_helpers import RaggedBatch
    from accvlab.lane_helpers.polyline import interpolate, interpolate_var_size_batch, lengths, lengths_var_size_batch

    rng = np.random.default_rng(9930 + seed)
    for case in range(6):
        b, p_max, q_max, dims = int(rng.integers(1, 6)), int(rng.choice([1, 2, 6, 40])), int(rng.choice([1, 5, 70])), int(rng.choice([2, 3]))
        dt = np.float32 if (seed + case) % 2 else np.float64
        relative = bool(rng.integers(0, 2))
        pts = (rng.random((b, p_max, dims)) * 4.0).astype(dt)
        for i in range(b):
            for j in range(1, p_max):
                if rng.random() < 0.15:
                    pts[i, j] = pts[i, j - 1]
        dist = ((rng.random((b, q_max)) * 1.4 - 0.2) * (1.0 if relative else 6.0)).astype(dt)
        tol = 4e-5 if dt == np.float32 else 1e-12
        got = interpolate(torch.from_numpy(pts), torch.from_numpy(dist), relative=relative).numpy()
        ln = lengths(torch.from_numpy(pts)).numpy()
        for i in range(b):
            assert np.allclose(got[i], oracle_lane.sample(pts[i], dist[i], relative), atol=tol, rtol=0), f"fixed {seed}/{case}/{i}"
            assert abs(ln[i] - oracle_lane.length(pts[i])) <= tol
        n_pts, n_q = rng.integers(0, p_max + 1, size=b), rng.integers(0, q_max + 1, size=b)
        rg = interpolate_var_size_batch(RaggedBatch(torch.from_numpy(pts), sample_sizes=torch.from_numpy(n_pts)),
                                        RaggedBatch(torch.from_numpy(dist), sample_sizes=torch.from_numpy(n_q)), relative=relative)
        rl = lengths_var_size_batch(RaggedBatch(torch.from_numpy(pts), sample_sizes=torch.from_numpy(n_pts))).numpy()
        for i in range(b):
            want = oracle_lane.sample(pts[i, : n_pts[i]], dist[i, : n_q[i]], relative)
            have = rg.tensor[i, : n_q[i]].numpy()
            assert np.array_equal(np.isnan(have), np.isnan(want)) and np.allclose(have, want, atol=tol, rtol=0, equal_nan=True), \
                f"ragged {seed}/{case}/{i}"
            wl_ = oracle_lane.length(pts[i, : n_pts[i]])
            assert (np.isnan(wl_) and np.isnan(rl[i])) or abs(rl[i] - wl_) <= tol


@pytest.mark.parametrize("seed", _seeds(8))
def test_masked_reductions_against_the_list_model(seed):
    """sum_over_targets / average_over_targets (empty samples: 0 or NaN) / apply_mask_to_tensor / squeeze_except_batch_and_sample
    over random batch shapes and positions of the non-uniform dimension, against per-sample sums of the valid entries"""
    from accvlab.batching_helpers import apply_mask_to_tensor, average_over_targets, squeeze_except_batch_and_sample, sum_over_targets

    rng = np.random.default_rng(9700 + seed)
    for case in range(8):
        rbatch, nb, nu_pos = _random_ragged(rng)
        leaves = _model(rbatch)
        shape = tuple(rbatch.batch_shape)
        data_shape = list(rbatch.tensor.shape[nb:])
        del data_shape[nu_pos]
        want_sum = torch.stack([l.sum(dim=nu_pos) for l in leaves]).reshape(shape + tuple(data_shape))
        assert torch.allclose(sum_over_targets(rbatch), want_sum, rtol=1e-12, atol=1e-12), f"sum {seed}/{case}"
        sizes = rbatch.sample_sizes.reshape(shape + (1,) * len(data_shape)).to(torch.float64)
        # the average first SWAPS the non-uniform dimension with the first data dimension (as the reference does,
        # batched_processing_py.py:35-37): its data dimensions come out in that swapped order
        swapped = torch.stack([l.transpose(0, nu_pos).sum(dim=0) for l in leaves])
        want_avg_sum = swapped.reshape(shape + tuple(swapped.shape[1:]))
        avg = average_over_targets(rbatch)
        assert torch.allclose(avg, torch.where(sizes > 0, want_avg_sum / sizes.clamp(min=1), torch.zeros_like(want_avg_sum)),
                              rtol=1e-12, atol=1e-12), f"average {seed}/{case}"
        avg_nan = average_over_targets(rbatch, nans_to_zero=False)
        assert torch.equal(torch.isnan(avg_nan), (sizes == 0).expand(avg_nan.shape))
        # a mask over the leading dimensions, broadcast over the rest
        t = rbatch.tensor
        lead = int(rng.integers(1, t.dim() + 1))
        mask = torch.from_numpy(rng.random(tuple(t.shape[:lead])) < 0.5)
        masked = apply_mask_to_tensor(t, mask, -2.0)
        m = mask.reshape(tuple(mask.shape) + (1,) * (t.dim() - lead)).expand(t.shape)
        assert torch.equal(masked, torch.where(m, t, torch.full_like(t, -2.0)))
        # squeeze: data dimensions of extent 1 go, batch and non-uniform dimensions stay
        sq = squeeze_except_batch_and_sample(rbatch)
        keep = [i for i in range(t.dim()) if i < nb or i == rbatch.non_uniform_dim or t.shape[i] != 1]
        assert tuple(sq.tensor.shape) == tuple(t.shape[i] for i in keep) and sq.non_uniform_dim == keep.index(rbatch.non_uniform_dim)
        assert torch.equal(sq.tensor, t.reshape(sq.tensor.shape)) and torch.equal(sq.sample_sizes, rbatch.sample_sizes)
