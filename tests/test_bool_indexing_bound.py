"""Sync-free compaction (extension): ``max_sample_size=`` on batched_bool_indexing / get_compact_lists /
get_indices_from_mask gives the result a caller-known width instead of reading the largest count back from the device
(the one host synchronisation of these operators decides an output shape,
batched_bool_indexing.py:198 / batched_processing_py.py:245-246 of the reference)."""
import pytest
import torch

from accvlab.batching_helpers import (RaggedBatch, batched_bool_indexing, get_compact_lists, get_indices_from_mask)


def _case(dev, seed=0, b=6, m=20, d=3):
    g = torch.Generator().manual_seed(seed)
    data = torch.randn(b, m, d, generator=g).to(dev)
    mask = (torch.rand(b, m, generator=g) < 0.4).to(dev)
    return data, mask


def _check(dev):
    data, mask = _case(dev)
    ref = batched_bool_indexing(data, mask)                      # reference behaviour (one read-back)
    longest = ref.tensor.shape[1]
    for bound in (longest, longest + 5, 20, 100):
        got = batched_bool_indexing(data, mask, max_sample_size=bound)
        w = min(bound, 20)
        assert got.tensor.shape == (6, w, 3)
        assert torch.equal(got.sample_sizes, ref.sample_sizes)
        assert torch.equal(got.tensor[:, :longest], ref.tensor) and (got.tensor[:, longest:] == 0).all()
    # a bound below the true maximum drops the tail of the longer samples
    small = max(longest - 2, 0)
    got = batched_bool_indexing(data, mask, max_sample_size=small)
    assert got.tensor.shape[1] == small
    assert torch.equal(got.sample_sizes, ref.sample_sizes.clamp(max=small))
    keep = torch.arange(small, device=dev).unsqueeze(0) < got.sample_sizes.unsqueeze(1)
    assert torch.equal(got.tensor[keep], ref.tensor[:, :small][keep])
    # the other entry points share the switch
    idx_ref = get_indices_from_mask(mask)
    idx = get_indices_from_mask(mask, max_sample_size=longest + 3)
    assert idx.tensor.shape[1] == longest + 3 and torch.equal(idx.tensor[:, :longest], idx_ref.tensor)
    a, b_, other = get_compact_lists(mask, [data, data[..., 0], "x"], max_sample_size=longest + 1)
    assert a.tensor.shape[1] == longest + 1 and b_.tensor.shape[1] == longest + 1 and other == "x"
    assert torch.equal(a.tensor[:, :longest], ref.tensor)
    # ragged mask input
    rmask = RaggedBatch(mask, sample_sizes=torch.tensor([20, 3, 0, 10, 20, 7], device=dev))
    r_ref = batched_bool_indexing(data, rmask)
    r_got = batched_bool_indexing(data, rmask, max_sample_size=20)
    assert torch.equal(r_got.sample_sizes, r_ref.sample_sizes)
    assert torch.equal(r_got.tensor[:, :r_ref.tensor.shape[1]], r_ref.tensor)


def test_bound_cpu():
    _check(torch.device("cpu"))


@pytest.mark.gpu
def test_bound_gpu_and_graph_capture():
    dev = torch.device("cuda", 0)
    _check(dev)
    # with a bound nothing synchronises: the whole compaction can be captured into one hipGraph and replayed
    data, mask = _case(dev, seed=1)
    static_data, static_mask = data.clone(), mask.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        batched_bool_indexing(static_data, static_mask, max_sample_size=12)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = batched_bool_indexing(static_data, static_mask, max_sample_size=12)
    for seed in (2, 3):
        d2, m2 = _case(dev, seed=seed)
        static_data.copy_(d2)
        static_mask.copy_(m2)
        graph.replay()
        torch.cuda.synchronize()
        want = batched_bool_indexing(d2, m2, max_sample_size=12)
        assert torch.equal(out.tensor, want.tensor) and torch.equal(out.sample_sizes, want.sample_sizes)


@pytest.mark.gpu
@pytest.mark.parametrize("b,m", [(5, 64), (3, 513), (7, 900), (2, 4096), (3, 4097), (2, 70001), (1, 1), (2, 8192), (3, 20000),
                                 (8, 65536), (1, 8191)])
@pytest.mark.parametrize("ragged", [False, True])
def test_mask_to_indices_all_kernel_variants(b, m, ragged):
    # widths <= 512: a wave per row; <= 4096: 4 waves per row; above: 16 waves per row; >= 8192 with few rows: 4096-byte
    # segments, one workgroup each, two passes (16-byte loads when the row is aligned, byte loads otherwise) — all order preserving
    from accvlab.batching_helpers import batched_indexing_access_cuda as ext

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(b * 100003 + m)
    mask = (torch.rand(b, m, generator=g) < 0.37)
    valid = torch.randint(0, m + 1, (b,), generator=g) if ragged else None
    idx, sizes = ext.mask_to_indices(mask.to(dev), valid.to(dev) if ragged else None)
    idx, sizes = idx.cpu(), sizes.cpu()
    for i in range(b):
        row = mask[i].clone()
        if ragged:
            row[int(valid[i]):] = False
        want = torch.nonzero(row).flatten()
        assert int(sizes[i]) == want.numel()
        assert torch.equal(idx[i, :want.numel()], want) and bool((idx[i, want.numel():] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,density", [(8, 65536, 0.1), (2, 262144, 0.5), (1, 8192, 0.0), (4, 12288, 1.0), (3, 16384, 0.4), (2, 9001, 0.3), (5, 16383, 0.7),
                                         (16, 70001, 0.03),
                                         (64, 65536, 0.2), (1, 1048576, 0.01)])
def test_segmented_mask_to_indices_one_pass_two_pass_and_graph_replay(b, m, density):
    """few, very wide rows: ONE launch
    for rows of up to 4 segments (every segment workgroup counts its whole row itself), two launches beyond — same indices,
    call after call, and from a replayed graph"""
    from accvlab.batching_helpers import batched_indexing_access_cuda as ext

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(b * 7 + m)

    def want(mask, valid):
        out = torch.zeros(mask.shape, dtype=torch.int64)
        sizes = torch.zeros(mask.shape[0], dtype=torch.int64)
        for i in range(mask.shape[0]):
            row = mask[i].clone()
            if valid is not None:
                row[int(valid[i]):] = False
            nz = torch.nonzero(row).flatten()
            out[i, :nz.numel()] = nz
            sizes[i] = nz.numel()
        return out, sizes

    for rep in range(3):     # repeated calls recycle the allocator's workspace block: stale words of older tags must not count
        mask = torch.rand(b, m, generator=g) < density
        valid = torch.randint(0, m + 1, (b,), generator=g) if rep == 1 else None
        idx, sizes = ext.mask_to_indices(mask.to(dev), valid.to(dev) if valid is not None else None)
        w_idx, w_sizes = want(mask, valid)
        assert torch.equal(sizes.cpu(), w_sizes) and torch.equal(idx.cpu(), w_idx)
    # captured: the launch(es) must be replayable with new mask contents (the tagged one-pass path is not taken there)
    static = (torch.rand(b, m, generator=g) < density).to(dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ext.mask_to_indices(static)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_idx, g_sizes = ext.mask_to_indices(static)
    for _ in range(2):
        fresh = torch.rand(b, m, generator=g) < max(density, 0.05)
        static.copy_(fresh.to(dev))
        graph.replay()
        torch.cuda.synchronize()
        w_idx, w_sizes = want(fresh, None)
        assert torch.equal(g_sizes.cpu(), w_sizes) and torch.equal(g_idx.cpu(), w_idx)
