"""Direct GPU tests of the reduction / masking helpers of batching_helpers against the oracle (SURVEY §8 row 21:
average_over_targets, sum_over_targets, apply_mask_to_tensor, squeeze_except_batch_and_sample —
packages/batching_helpers/accvlab/batching_helpers/batched_processing_py.py:23-174).  On the GPU these run the HIP pad-fill
kernel (accv_ragged_pad_fill) followed by a torch reduction.  Oracle: oracle.h2.pad_fill (pinned by the reference's
pad-fill literals) + numpy sums; the padding holds NaN / junk, which must never reach the result."""
import numpy as np
import pytest
import torch

from oracle import h2 as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(seed, batch_shape, width, inner, dtype, nu_last=False):
    g = np.random.default_rng(seed)
    shape = tuple(batch_shape) + (width,) + tuple(inner)
    data = g.normal(size=shape).astype(np.float64)
    sizes = g.integers(0, width + 1, size=batch_shape).astype(np.int64)
    sizes.reshape(-1)[0] = width
    if sizes.size > 1:
        sizes.reshape(-1)[1] = 0                                       # an empty sample
    valid = np.arange(width).reshape((1,) * len(batch_shape) + (width,)) < sizes[..., None]
    junk = np.where(valid.reshape(valid.shape + (1,) * len(inner)), data, np.nan)   # NaN in every padded entry
    return junk.astype(dtype), sizes, valid


@pytest.mark.parametrize("dtype,tol", [(np.float32, 1e-5), (np.float64, 1e-12)])
@pytest.mark.parametrize("batch_shape,width,inner", [((6,), 9, (4,)), ((2, 3), 5, (2, 3)), ((4,), 17, ()), ((1,), 1, (1,))])
def test_sum_and_average_over_targets_vs_oracle(dtype, tol, batch_shape, width, inner):
    import accvlab.batching_helpers as bh

    data, sizes, valid = _case(len(inner) + width, batch_shape, width, inner, dtype)
    nb = len(batch_shape)
    rb = bh.RaggedBatch(torch.from_numpy(data).to(DEV), sample_sizes=torch.from_numpy(sizes).to(DEV))
    flat = data.reshape((-1, width) + tuple(inner))
    zeroed = oracle.pad_fill(flat, sizes.reshape(-1), 0.0).reshape(data.shape).astype(np.float64)
    want_sum = zeroed.sum(axis=nb)
    got_sum = bh.sum_over_targets(rb)
    assert got_sum.dtype == rb.tensor.dtype and tuple(got_sum.shape) == want_sum.shape
    assert float(np.abs(got_sum.cpu().numpy() - want_sum).max()) <= tol * max(1.0, float(np.abs(want_sum).max()))
    cnt = sizes.reshape(sizes.shape + (1,) * len(inner)).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        want_avg = np.nan_to_num(want_sum / cnt, nan=0.0, posinf=0.0, neginf=0.0)
    got_avg = bh.average_over_targets(rb)
    assert float(np.abs(got_avg.cpu().numpy() - want_avg).max()) <= tol * max(1.0, float(np.abs(want_avg).max()))
    raw = bh.average_over_targets(rb, nans_to_zero=False).cpu().numpy()
    empty = (sizes == 0)
    if empty.any():
        assert np.isnan(raw[empty]).all()                                # 0 / 0 for samples without targets
    # the input batch is untouched (out-of-place masking): its padding still holds the NaNs
    assert bool(torch.isnan(rb.tensor).any()) == bool((~valid).any())


def test_sum_over_targets_transposed_non_uniform_dim_and_gradient():
    import accvlab.batching_helpers as bh

    data, sizes, valid = _case(5, (4,), 6, (3,), np.float32)
    t = torch.from_numpy(np.nan_to_num(data, nan=7.0)).to(DEV).requires_grad_(True)
    rb = bh.RaggedBatch(t, sample_sizes=torch.from_numpy(sizes).to(DEV))
    tr = rb.get_non_uniform_dimension_transposed_to(2)                  # [B, 3, 6], non-uniform dim 2
    s1, s2 = bh.sum_over_targets(rb), bh.sum_over_targets(tr)
    assert torch.allclose(s1, s2, atol=1e-6)
    s1.sum().backward()
    assert torch.equal(t.grad, torch.from_numpy(valid[..., None].repeat(3, -1)).to(DEV).to(torch.float32))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int64])
def test_apply_mask_to_tensor_and_squeeze_on_gpu(dtype):
    import accvlab.batching_helpers as bh

    g = torch.Generator().manual_seed(1)
    data = (torch.rand(3, 5, 2, 4, generator=g) * 100).to(dtype).to(DEV)
    mask = (torch.rand(3, 5, generator=g) < 0.5).to(DEV)
    got = bh.apply_mask_to_tensor(data, mask, 9.0)
    want = data.clone()
    want[~mask] = 9
    assert torch.equal(got, want) and got.data_ptr() != data.data_ptr()
    assert torch.equal(bh.apply_mask_to_tensor(data, torch.ones_like(mask)), data)
    rb = bh.RaggedBatch(torch.rand(3, 1, 5, 1, 4, generator=g).to(DEV), sample_sizes=torch.tensor([5, 2, 0], device=DEV),
                        non_uniform_dim=2)
    sq = bh.squeeze_except_batch_and_sample(rb)
    assert tuple(sq.shape) == (3, 5, 4) and sq.non_uniform_dim == 1 and torch.equal(sq.sample_sizes, rb.sample_sizes)
    assert torch.equal(sq.tensor, rb.tensor.reshape(3, 5, 4))
    plain = bh.squeeze_except_batch_and_sample(torch.zeros(2, 1, 1, 3, 1, device=DEV))
    assert tuple(plain.shape) == (2, 1, 3)
