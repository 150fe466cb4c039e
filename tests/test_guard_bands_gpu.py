"""Out-of-bounds discipline of the hand-written kernels: every output lives INSIDE a larger buffer filled with a
sentinel; after the call the margins must be untouched (odd extents: W not a multiple of 4 / 128, H not a multiple of
16, unaligned base pointers, ragged row widths that are not a multiple of the vector width)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SENT = -12345.0


def _inside(shape, dtype=torch.float32, pad=257, offset=0):
    """contiguous tensor of `shape` carved out of a sentinel-filled flat buffer, `offset` elements off alignment"""
    n = int(np.prod(shape))
    buf = torch.full((n + 2 * pad + offset,), SENT, dtype=dtype, device=DEV)
    view = buf[pad + offset: pad + offset + n].view(shape)
    return buf, view, pad + offset, n


def _margins_clean(buf, start, n):
    return bool((buf[:start] == SENT).all()) and bool((buf[start + n:] == SENT).all())


@pytest.mark.parametrize("H,W", [(16, 128), (17, 129), (33, 100), (5, 7), (64, 260), (1, 1), (50, 1924)])
@pytest.mark.parametrize("offset", [0, 1, 3])
@pytest.mark.parametrize("mode", ["clear", "inplace", "small", "flat", "classwise"])
def test_heatmap_kernels_stay_inside_the_map(H, W, offset, mode):
    from accvlab.draw_heatmap import draw_heatmap, draw_heatmap_batched

    B, N, C = 3, 9, 2
    g = torch.Generator().manual_seed(H * 1000 + W)
    centers = torch.stack([torch.randint(-4, W + 4, (B, N), generator=g),
                           torch.randint(-4, H + 4, (B, N), generator=g)], -1).to(torch.int32).to(DEV)
    radii = torch.randint(0, 40, (B, N), generator=g).to(torch.int32).to(DEV)
    sizes = torch.tensor([N, 0, 4], device=DEV)
    rb = lambda t: SimpleNamespace(tensor=t, sample_sizes=sizes)  # noqa: E731
    shape = (B, C, H, W) if mode == "classwise" else (B, H, W)
    buf, hm, start, n = _inside(shape, offset=offset)
    hm.fill_(0.0)
    if mode == "flat":
        idx = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(N)
        draw_heatmap(hm, centers.reshape(-1, 2).contiguous(), radii.reshape(-1).contiguous(), idx)
    elif mode == "classwise":
        labels = torch.randint(0, C, (B, N), generator=g).to(torch.int32).to(DEV)
        draw_heatmap_batched(hm, rb(centers), rb(radii), 6.0, 1.0, rb(labels))
    else:
        draw_heatmap_batched(hm, rb(centers), rb(radii), 6.0, 1.0, clear=mode != "inplace", small_radii=mode == "small")
    torch.cuda.synchronize()
    assert _margins_clean(buf, start, n), f"{mode} {H}x{W} offset {offset}: wrote outside the heat-map"
    assert float(hm.max()) <= 1.0 + 1e-6 and float(hm.min()) >= 0.0


@pytest.mark.parametrize("row", [1, 3, 4, 6, 16, 33])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int64])
def test_ragged_kernels_stay_inside_their_outputs(row, dtype):
    from accvlab.batching_helpers import RaggedBatch, batched_indexing_access, batched_inverse_indexing_access
    from accvlab.batching_helpers import batched_indexing_access_cuda as ext

    b, n, k = 4, 11, 7
    g = torch.Generator().manual_seed(row)
    data = (torch.randn(b, n, row, generator=g) * 10).to(dtype).to(DEV)
    idx = torch.stack([torch.randperm(n, generator=g)[:k] for _ in range(b)]).to(DEV)
    counts = torch.tensor([k, 0, 3, 5], device=DEV)
    # gather / scatter through the extension entry points that write into caller-provided tensors
    buf, out, start, m = _inside((b, k, row), dtype=dtype)
    out.zero_()
    ext.gather_rows(data, idx, counts, k, out)
    torch.cuda.synchronize()
    assert _margins_clean(buf, start, m)
    buf2, dst, start2, m2 = _inside((b, n, row), dtype=dtype)
    dst.zero_()
    ext.scatter_rows(out, idx, counts, k, dst)
    torch.cuda.synchronize()
    assert _margins_clean(buf2, start2, m2)
    for i in range(b):
        c = int(counts[i])
        assert torch.equal(dst[i, idx[i, :c]], data[i, idx[i, :c]])
    # pad fill in place
    buf3, padded, start3, m3 = _inside((b, k, row), dtype=dtype)
    padded.copy_(out)
    RaggedBatch(padded, sample_sizes=counts).set_padded_to(7)
    torch.cuda.synchronize()
    assert _margins_clean(buf3, start3, m3)
    # allocating operators: results equal the per-sample formulation (their outputs are fresh tensors)
    got = batched_indexing_access(data, RaggedBatch(idx, sample_sizes=counts), 0)
    back = batched_inverse_indexing_access(got, RaggedBatch(idx, sample_sizes=counts), n, 0)
    for i in range(b):
        c = int(counts[i])
        assert torch.equal(got.tensor[i, :c], data[i, idx[i, :c]]) and bool((got.tensor[i, c:] == 0).all())
        assert torch.equal(back[i, idx[i, :c]], data[i, idx[i, :c]])


@pytest.mark.parametrize("P,Q,D", [(2, 1, 2), (13, 9, 3), (70, 130, 2)])
def test_polyline_kernel_stays_inside_its_output(P, Q, D):
    from accvlab import _amd_native as nat

    b = 5
    g = torch.Generator().manual_seed(P)
    pts = torch.randn(b, P, D, generator=g).cumsum(1).to(DEV)
    dist = (torch.rand(b, Q, generator=g) * 5).to(DEV)
    buf, out, start, n = _inside((b, Q, D))
    lib = nat.lib()
    sb = lib.accv_polyline_scratch_bytes(b, P, 0)
    scratch = torch.empty(max(sb, 1), dtype=torch.uint8, device=DEV)
    nat.check(lib.accv_polyline_sample(pts.data_ptr(), dist.data_ptr(), None, None, out.data_ptr(), None, b, P, Q, D, 0, 0,
                                       0, scratch.data_ptr(), sb, nat.stream_ptr(DEV)), "polyline")
    torch.cuda.synchronize()
    assert _margins_clean(buf, start, n)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("clear", [True, False])
def test_multiscale_kernels_stay_inside_their_maps(clear):
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale

    b = 3
    g = torch.Generator().manual_seed(11)
    cs, bs = [], []
    for _ in range(b):
        n = int(torch.randint(0, 12, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([500.0, 300.0])
        half = torch.rand(n, 4, generator=g) * 60
        cs.append(c)
        bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
    crb = combine_data(cs, device=DEV)
    brb = combine_data(bs, device=DEV, other_with_same_sample_sizes=crb)
    lanes = (torch.rand(b, 4, 9, 2, generator=g) * torch.tensor([520.0, 320.0]) - 10).to(DEV)   # partly outside
    shapes = [(b, 75, 128), (b, 37, 64), (b, 19, 32), (b, 9, 16)]       # heights not multiples of 16, widths of 4
    strides = (4.0, 8.0, 16.0, 32.0)
    bufs = [_inside(s, pad=256) for s in shapes]                        # 256-float pad keeps the 16-byte alignment
    for _, view, _, _ in bufs:
        view.fill_(0.1)
    maps = [v for _, v, _, _ in bufs]
    draw_heatmap_multiscale(maps, crb, brb, strides, clear=clear)
    draw_polylines_multiscale(maps, lanes, 70, 2, strides, clear=False)
    torch.cuda.synchronize()
    for (buf, view, start, n), s in zip(bufs, shapes):
        assert _margins_clean(buf, start, n), f"multi-scale kernels wrote outside the {s} map"
        assert torch.isfinite(view).all()


@pytest.mark.parametrize("count,offset", [(0, 0), (1, 0), (3, 1), (4, 0), (1023, 3), (1 << 20, 0), ((1 << 20) + 5, 2)])
def test_fill_f32_fills_exactly_its_range(count, offset):
    from accvlab import _amd_native as nat

    buf, view, start, n = _inside((max(count, 1),), offset=offset)
    view.fill_(1.0)
    nat.check(nat.lib().accv_fill_f32(view.data_ptr(), count, 2.5, nat.stream_ptr(DEV)), "fill")
    torch.cuda.synchronize()
    assert _margins_clean(buf, start, n)
    assert bool((view[:count] == 2.5).all()) and bool((view[count:] == 1.0).all())
