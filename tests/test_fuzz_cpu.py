"""Seeded random sweeps of the CPU side of batching_helpers (config C0 runs here: torch-CPU tensors, the host extension and its
python twin) against the oracle: pack / mask / split round trips over random sample lengths, inner shapes and dtypes incl. empty
samples and nested batch dimensions; compaction and pad fill of CPU tensors."""
import numpy as np
import pytest
import torch

from oracle import h2 as oracle_h2

_DTYPES = [torch.float32, torch.float64, torch.int64, torch.int32, torch.float16, torch.uint8, torch.bool]


def _rand(shape, dt, rng):
    a = rng.random(shape) * 60
    return torch.from_numpy(a > 30) if dt == torch.bool else torch.from_numpy(a).to(dt)


@pytest.mark.parametrize("seed", range(10))
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


@pytest.mark.parametrize("seed", range(6))
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


@pytest.mark.parametrize("seed", range(6))
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
