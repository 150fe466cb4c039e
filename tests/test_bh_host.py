"""The C++ per-sample loops of batching_helpers (csrc_host/bh_host.cpp: combine_data pack, RaggedBatch.split views)
must be indistinguishable from the python implementation they shortcut (which mirrors
batched_processing_py.py:410-427 and ragged_batch.py:870-934 of the reference)."""
import pytest
import torch

from accvlab.batching_helpers import RaggedBatch, combine_data, packing, ragged


def _both(fn):
    """run fn with the C++ loops and with the python fallback"""
    assert packing._bh is not None and ragged._bh is not None, "the batching_helpers host extension was not built"
    fast = fn()
    saved = packing._bh, ragged._bh
    packing._bh = ragged._bh = None
    try:
        slow = fn()
    finally:
        packing._bh, ragged._bh = saved
    return fast, slow


def _same_rb(a: RaggedBatch, b: RaggedBatch):
    assert a.tensor.dtype == b.tensor.dtype and a.tensor.shape == b.tensor.shape
    assert torch.equal(a.tensor, b.tensor)
    assert torch.equal(a.sample_sizes, b.sample_sizes) and a.sample_sizes.dtype == b.sample_sizes.dtype
    assert torch.equal(a.mask, b.mask)
    assert a.non_uniform_dim == b.non_uniform_dim and a.num_batch_dims == b.num_batch_dims


def _samples(seed, n=17, inner=(4,), dtype=torch.float32, with_empty=True):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        k = int(torch.randint(0 if with_empty else 1, 9, (1,), generator=g))
        out.append((torch.randn(k, *inner, generator=g) * 10).to(dtype))
    return out


@pytest.mark.parametrize("inner,dtype", [((4,), torch.float32), ((), torch.int64), ((2, 3), torch.float64),
                                         ((5,), torch.float16), ((1,), torch.bool)])
def test_pack_matches_python(inner, dtype):
    data = _samples(1, inner=inner, dtype=dtype)
    fast, slow = _both(lambda: combine_data(data))
    _same_rb(fast, slow)


def test_pack_nested_tuple_noncontiguous_and_shared_sizes():
    data = _samples(2, n=12, inner=(6,), with_empty=False)
    data[3] = data[3][:, ::2].repeat(1, 2)[:, ::1].t().contiguous().t()      # non-contiguous leaf, same shape
    assert not data[3].is_contiguous()
    nested = [tuple(data[:5]), [data[5:9], (data[9], data[10])], data[11]]
    fast, slow = _both(lambda: combine_data(nested))
    _same_rb(fast, slow)
    other = combine_data([torch.zeros(t.shape[0], 2) for t in data])
    fast, slow = _both(lambda: combine_data(nested, other_with_same_sample_sizes=other))
    _same_rb(fast, slow)
    assert fast.sample_sizes is other.sample_sizes or torch.equal(fast.sample_sizes, other.sample_sizes)


def test_cases_the_cpp_path_declines_still_work():
    # mixed dtypes (converted to the first non-empty sample's dtype), gradients, a device argument, all-empty input
    mixed = [torch.ones(2, 3), torch.ones(1, 3, dtype=torch.float64)]
    fast, slow = _both(lambda: combine_data(mixed))
    _same_rb(fast, slow)
    leaf = torch.ones(2, 3, requires_grad=True)
    rb = combine_data([leaf, torch.zeros(1, 3)])
    assert rb.tensor.requires_grad
    rb.tensor.sum().backward()
    assert torch.equal(leaf.grad, torch.ones(2, 3))
    fast, slow = _both(lambda: combine_data(_samples(3), device="cpu"))
    _same_rb(fast, slow)
    fast, slow = _both(lambda: combine_data([torch.zeros(0, 4), torch.zeros(0, 4)]))
    _same_rb(fast, slow)
    with pytest.raises(AssertionError):
        combine_data([torch.zeros(1, 2), "not a tensor"])


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("nested_batch", [False, True])
def test_split_matches_python(transposed, nested_batch):
    g = torch.Generator().manual_seed(5)
    if nested_batch:
        t = torch.randn(2, 3, 7, 4, generator=g)
        sizes = torch.randint(0, 8, (2, 3), generator=g)
        rb = RaggedBatch(t, sample_sizes=sizes, non_uniform_dim=2)
    else:
        t = torch.randn(6, 7, 4, generator=g)
        rb = RaggedBatch(t, sample_sizes=torch.randint(0, 8, (6,), generator=g))
    if transposed:
        rb = rb.get_non_uniform_dimension_transposed_to(rb.non_uniform_dim + 1)

    def flat(x):
        return [x] if isinstance(x, torch.Tensor) else [y for el in x for y in flat(el)]

    fast, slow = _both(lambda: rb.split())
    ff, fs = flat(fast), flat(slow)
    assert len(ff) == len(fs) == rb.sample_sizes.numel()
    for a, b in zip(ff, fs):
        assert a.shape == b.shape and a.stride() == b.stride() and torch.equal(a, b)
        assert a.data_ptr() == b.data_ptr()                    # both are views of the batch tensor
    # writes through a split view land in the batch tensor
    if ff[0].numel():
        ff[0].fill_(123.0)
        assert (rb.tensor == 123.0).any()


def test_split_of_a_tensor_that_requires_grad_stays_differentiable():
    t = torch.randn(3, 5, 2, requires_grad=True)
    rb = RaggedBatch(t * 1.0, sample_sizes=torch.tensor([5, 2, 0]))
    parts = rb.split()
    (parts[0].sum() + 2 * parts[1].sum()).backward()
    want = torch.zeros(3, 5, 2)
    want[0] = 1.0
    want[1, :2] = 2.0
    assert torch.equal(t.grad, want)


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["cpu", "cuda"])
@pytest.mark.parametrize("inner,dtype", [((4,), torch.float32), ((), torch.int64), ((2, 3), torch.float16)])
def test_gpu_target_fast_paths_match_python(source, inner, dtype):
    dev = torch.device("cuda", 0)
    data = _samples(7, n=33, inner=inner, dtype=dtype)
    if source == "cuda":
        data = [t.to(dev) for t in data]
    nested = [data[:10], (data[10:20], [data[20]]), data[21:]]
    for arg in (data, nested):
        fast, slow = _both(lambda: combine_data(arg, device=dev))
        assert fast.tensor.device == dev and fast.sample_sizes.device == dev
        _same_rb(fast, slow)
    if source == "cuda":
        fast, slow = _both(lambda: combine_data(data))          # device inferred from the samples
        _same_rb(fast, slow)
    # a batch above the direct-transfer limit takes the flat + pack-kernel route
    big = [torch.randn(int(n), 512).to(dtype if dtype.is_floating_point else torch.float32)
           for n in torch.randint(1, 40, (64,))]
    if source == "cuda":
        big = [t.to(dev) for t in big]
    fast, slow = _both(lambda: combine_data(big, device=dev))
    _same_rb(fast, slow)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_gpu_split_views_alias_the_batch():
    dev = torch.device("cuda", 0)
    rb = RaggedBatch(torch.randn(5, 6, 3, device=dev), sample_sizes=torch.tensor([6, 0, 2, 5, 1], device=dev))
    fast, slow = _both(lambda: rb.split())
    for a, b in zip(fast, slow):
        assert a.shape == b.shape and a.data_ptr() == b.data_ptr() and torch.equal(a, b)


def test_split_views_share_the_version_counter_of_the_batch():
    # a sample returned by split() is a view of the padded tensor — an in-place write through it must bump
    # the batch tensor's version so autograd's "modified in place" check of anything that saved it still fires
    rb = RaggedBatch(torch.arange(24.0).reshape(3, 4, 2).clone(), sample_sizes=torch.tensor([4, 2, 1]))
    before = rb.tensor._version
    parts = rb.split()
    parts[1].mul_(2.0)
    assert rb.tensor._version > before
    # and the check itself: w * tensor saves the tensor; editing a split view afterwards must make backward fail
    w = torch.ones(3, 4, 2, requires_grad=True)
    rb2 = RaggedBatch(torch.rand(3, 4, 2), sample_sizes=torch.tensor([4, 2, 1]))
    y = (w * rb2.tensor).sum()
    rb2.split()[0].zero_()
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.backward()


@pytest.mark.gpu
def test_split_uses_the_host_copy_of_the_sizes_and_detects_edits():
    """combine_data knows the sample sizes on the host; split() of the result (and of batches sharing its size tensor) must
    not read them back from the device — and must, as soon as the size tensor was edited in place"""
    from accvlab.batching_helpers import ragged

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    samples = [torch.rand(int(torch.randint(0, 9, (1,), generator=g)), 3, generator=g) for _ in range(12)]
    for src in (samples, [t.to(dev) for t in samples]):          # CPU samples -> GPU, and GPU samples
        rb = combine_data(src, device=dev)
        hit = getattr(rb.sample_sizes, "_accv_host_sizes", None)
        assert hit is not None and hit[0] == [t.shape[0] for t in samples]
        calls = []
        real = torch.Tensor.tolist
        torch.Tensor.tolist = lambda self: (calls.append(1), real(self))[1]
        try:
            parts = rb.split()
            other = rb.create_with_sample_sizes_like_self(rb.tensor * 2)      # shares the size tensor
            parts2 = other.split()
            tr = rb.get_non_uniform_dimension_transposed_to(2).split()
        finally:
            torch.Tensor.tolist = real
        assert calls == [], "split() read the sizes back although their host copy was valid"
        assert all(torch.equal(a.cpu(), b) for a, b in zip(parts, samples))
        assert all(torch.equal(a.cpu(), b * 2) for a, b in zip(parts2, samples))
        assert all(torch.equal(a.cpu(), b.t()) for a, b in zip(tr, samples))
        # an in-place edit of the sizes bumps their version: the host copy is dropped and the new sizes are used
        rb.sample_sizes.clamp_(max=1)
        edited = rb.split()
        assert [p.shape[0] for p in edited] == [min(1, t.shape[0]) for t in samples]
    assert ragged.host_sizes(torch.tensor([3, 1, 2])) == [3, 1, 2]


@pytest.mark.gpu
def test_native_mask_to_indices_agrees_with_the_python_formulation_and_declines_unusual_inputs():
    """mask_to_indices: checks + the three allocations + the launch in C++ (csrc_host/bh_host.cpp) — same tensors as the python
    formulation over the same C-ABI entry point; anything unusual is declined and diagnosed by the python code"""
    from accvlab import _amd_native as nat
    from accvlab.batching_helpers import batched_indexing_access_cuda as ext

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    for b, w, ragged in ((3, 40, False), (5, 700, True), (2, 5000, False), (8, 65536, True), (1, 8192, False)):
        mask = (torch.rand(b, w, generator=g) < 0.3).to(dev)
        valid = torch.randint(0, w + 1, (b,), generator=g).to(dev) if ragged else None
        got = ext.mask_to_indices(mask, valid)
        saved = nat.NO_HOST_FASTPATH
        nat.NO_HOST_FASTPATH = True
        try:
            want = ext.mask_to_indices(mask, valid)
        finally:
            nat.NO_HOST_FASTPATH = saved
        assert got[0].dtype == torch.int64 and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # declined -> python path: non-bool masks are converted, wrong ranks / devices raise the usual errors
    idx, sizes = ext.mask_to_indices((torch.arange(12, device=dev).reshape(3, 4) % 3).to(torch.int32))
    assert sizes.tolist() == [2, 3, 3]
    with pytest.raises(RuntimeError):
        ext.mask_to_indices(torch.ones(4, dtype=torch.bool, device=dev))
    with pytest.raises(RuntimeError):
        ext.mask_to_indices(torch.ones(2, 4, dtype=torch.bool))


@pytest.mark.gpu
def test_sizes_rewritten_behind_the_version_counter():
    """a device size tensor that was merely READ once is read again by every split() (the read-back is not cached —
    `.data` writes, raw-pointer kernels and graph replays do not bump the version counter); a combine_data size tensor
    that is rewritten that way needs forget_host_sizes()"""
    from accvlab.batching_helpers import RaggedBatch, ragged

    dev = torch.device("cuda", 0)
    data = torch.arange(24.0, device=dev).reshape(3, 4, 2)
    sizes = torch.tensor([4, 2, 3], device=dev)
    rb = RaggedBatch(data, sample_sizes=sizes)
    assert [p.shape[0] for p in rb.split()] == [4, 2, 3]
    assert getattr(sizes, "_accv_host_sizes", None) is None          # nothing cached from a read-back
    sizes.data.copy_(torch.tensor([1, 0, 2], device=dev))            # invisible to the version counter
    assert [p.shape[0] for p in rb.split()] == [1, 0, 2]
    # combine_data's host copy survives such a write (documented): the caller drops it explicitly
    rb2 = combine_data([torch.rand(n, 2) for n in (3, 1, 2)], device=dev)
    rb2.sample_sizes.data.copy_(torch.tensor([1, 1, 1], device=dev))
    ragged.forget_host_sizes(rb2.sample_sizes)
    assert [p.shape[0] for p in rb2.split()] == [1, 1, 1]
    ragged.forget_host_sizes(rb2.sample_sizes)                       # idempotent


def test_native_cpu_mask_equals_the_python_formulation():
    """RaggedBatch.mask for host data goes through _bh_host.mask_cpu: (arange(n) < sizes[..., None]) for any batch shape,
    int32 / int64 sizes, zero widths and sizes outside [0, n]"""
    from accvlab.batching_helpers import RaggedBatch, ragged

    if ragged._bh is None or not hasattr(ragged._bh, "mask_cpu"):
        pytest.skip("host extension not built")
    g = torch.Generator().manual_seed(0)
    for shape, n in (((7,), 5), ((2, 3), 9), ((4,), 0), ((1, 1, 2), 3)):
        for dtype in (torch.int64, torch.int32):
            sizes = torch.randint(-2, n + 3, shape, generator=g).to(dtype)
            got = ragged._bh.mask_cpu(sizes, n)
            want = torch.arange(n) < sizes.unsqueeze(-1)
            assert got.dtype == torch.bool and got.shape == (*shape, n) and torch.equal(got, want)
    sizes = torch.tensor([3, 0, 5])
    rb = RaggedBatch(torch.zeros(3, 5, 2), sample_sizes=sizes)
    assert torch.equal(rb.mask, torch.arange(5) < sizes.unsqueeze(-1))
    assert torch.equal(ragged._bh.mask_cpu(sizes[::2], 4), torch.arange(4) < sizes[::2].unsqueeze(-1))     # non-contiguous sizes
