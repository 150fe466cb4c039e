"""draw_targets_multiscale (box maps + lane maps of one step; the polyline sampler rides in the box-map launch) — GPU parity.

Bit-identity against the two operators it replaces, `draw_heatmap_multiscale` and `draw_polylines_multiscale` (which are pinned
against the oracles in tests/test_multiscale_gpu.py and tests/test_lane_raster_gpu.py), and of the rider itself against the
stand-alone sampler launch (`sample_lanes` = accv_polyline_sample_boxes: samples and group boxes)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _same_samples(a, b):
    """bit-identical finite values and NaNs in the same places (the sign bit of a NaN sample is not part of the contract: it
    depends on which of two NaN operands the multiply-add hardware passes on, and a NaN sample is never drawn)"""
    na, nb = torch.isnan(a), torch.isnan(b)
    zero = torch.zeros((), device=a.device)
    return bool((na == nb).all()) and torch.equal(torch.where(na, zero, a).view(torch.int32), torch.where(nb, zero, b).view(torch.int32))


def _objects(b, n_max, sw, sh, seed):
    from accvlab.batching_helpers import RaggedBatch

    g = torch.Generator().manual_seed(seed)
    c = torch.rand(b, n_max, 2, generator=g) * torch.tensor([sw, sh])
    half = torch.rand(b, n_max, 2, generator=g) * 60 + 2
    boxes = torch.cat([c - half, c + half], -1)
    n = torch.randint(0, n_max + 1, (b,), generator=g)
    return RaggedBatch(c.to(DEV), sample_sizes=n.to(DEV)), RaggedBatch(boxes.to(DEV), sample_sizes=n.to(DEV))


def _polylines(b, l, p, sw, sh, seed, ragged):
    g = np.random.default_rng(seed)
    start = g.uniform([0, 0], [sw, sh], size=(b, l, 1, 2))
    steps = g.normal(0, 1, size=(b, l, p, 2)) * [sw / p / 2, sh / p / 2] + [sw / p / 3, -sh / p / 4]
    pts = torch.from_numpy((start + np.cumsum(steps, axis=2)).astype(np.float32)).to(DEV)
    npts = torch.from_numpy(g.integers(0, p + 1, size=(b, l)).astype(np.int64)).to(DEV) if ragged else None
    nlanes = torch.from_numpy(g.integers(0, l + 1, size=(b,)).astype(np.int32)).to(DEV) if ragged else None
    return pts, npts, nlanes


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("clear", [True, False])
@pytest.mark.parametrize("l,p,q,radius", [(8, 24, 256, 2), (3, 64, 64, 1), (5, 1, 128, 3), (6, 11, 192, 0), (2, 65, 128, 2),
                                          (4, 12, 100, 2), (2, 24, 256, 2)])
def test_targets_equal_the_two_operators(ragged, clear, l, p, q, radius):
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, draw_targets_multiscale

    b, sw, sh = 3, 1536.0, 864.0
    strides = (4.0, 8.0, 16.0)
    centers, boxes = _objects(b, 20, sw, sh, seed=l * 7 + p)
    pts, npts, nlanes = _polylines(b, l, p, sw, sh, seed=l * 100 + p + q, ragged=ragged)
    shapes = [(b, int(sh / s), int(sw / s)) for s in strides]
    base = [torch.rand(s_, generator=torch.Generator().manual_seed(i)).mul_(0.3).to(DEV) for i, s_ in enumerate(shapes)]
    box_a, lane_a = [t.clone() for t in base], [t.clone() for t in base]
    box_b, lane_b = [t.clone() for t in base], [t.clone() for t in base]
    draw_targets_multiscale(box_a, centers, boxes, strides, lane_a, pts, q, radius, None, 6.0, 0.9, num_points=npts,
                            num_lanes=nlanes, clear=clear)
    assert "splat_points_multi_kernel" in nat.last_dispatch() or p > 64 or q % 64      # (else: the per-scale / two-launch paths)
    draw_heatmap_multiscale(box_b, centers, boxes, strides, 6.0, 0.9, clear=clear)
    draw_polylines_multiscale(lane_b, pts, q, radius, strides, 6.0, 0.9, num_points=npts, num_lanes=nlanes, clear=clear)
    if (l, p, q) == (2, 24, 256):     # a sparse lane set: alone it takes the one-launch lane raster, inside the step the rider
        assert "lane_raster_multi_kernel" in nat.last_dispatch()
    for i in range(len(strides)):
        assert torch.equal(box_a[i], box_b[i]), f"box map {i}"
        assert torch.equal(lane_a[i], lane_b[i]), f"lane map {i}"


@pytest.mark.parametrize("counts_dtype", [None, torch.int32, torch.int64])
@pytest.mark.parametrize("l,p,q", [(8, 24, 256), (1, 64, 64), (7, 1, 64), (5, 2, 128), (3, 37, 320)])
def test_rider_writes_what_the_sampler_launch_writes(l, p, q, counts_dtype):
    """samples and group boxes of the wave-level sampler == accv_polyline_sample_boxes, bit for bit"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import lanes, ops, sample_lanes

    b, sw, sh = 4, 2048.0, 1024.0
    centers, boxes = _objects(b, 12, sw, sh, seed=3)
    pts, npts, _ = _polylines(b, l, p, sw, sh, seed=l * 10 + p, ragged=counts_dtype is not None)
    if npts is not None:
        npts = npts.to(counts_dtype)
        npts[0, 0] = 0                                   # an empty polyline: NaN samples, inverted boxes
        if p >= 2:
            npts[1, 0] = 1
    pts[2, l - 1, p // 2] = float("nan")                 # a NaN vertex: NaN from there on
    groups = b * l * q // 64
    work = torch.zeros(nat.lib().accv_draw_points_workspace_bytes(b, l * q), dtype=torch.uint8, device=DEV)
    job = lanes._SamplerJob(pts, npts, q, work)
    maps = [torch.zeros(b, 128, 256, device=DEV)]
    ops.draw_heatmap_multiscale(maps, centers, boxes, (8.0,), clear=True, _sampler_job=job)
    assert "splat_multi_sampler_kernel" in nat.last_dispatch()
    got_boxes = work[: groups * 16].view(torch.float32).view(groups, 4).clone()
    ref_boxes = torch.zeros(groups, 4, device=DEV)
    ref = sample_lanes(pts, q, num_points=npts, group_boxes_ptr=ref_boxes.data_ptr())
    torch.cuda.synchronize()
    assert _same_samples(job.samples, ref), "samples differ (bit pattern)"
    assert torch.equal(got_boxes, ref_boxes)
    # ... and the box maps of that launch are those of the plain launch
    plain = [torch.zeros(b, 128, 256, device=DEV)]
    ops.draw_heatmap_multiscale(plain, centers, boxes, (8.0,), clear=True)
    assert "splat_multi_kernel" in nat.last_dispatch()
    assert torch.equal(maps[0], plain[0])


def test_rider_alone_and_argument_checks():
    from accvlab import _amd_native as nat
    from accvlab.batching_helpers import RaggedBatch
    from accvlab.draw_heatmap import lanes, ops, sample_lanes

    b, l, p, q = 2, 3, 9, 64
    pts, _, _ = _polylines(b, l, p, 500.0, 300.0, seed=1, ragged=False)
    work = torch.zeros(nat.lib().accv_draw_points_workspace_bytes(b, l * q), dtype=torch.uint8, device=DEV)
    job = lanes._SamplerJob(pts, None, q, work)
    # no objects at all and no clear: the box maps have nothing to launch, the sampler still runs
    empty_c = RaggedBatch(torch.zeros(b, 0, 2, device=DEV), sample_sizes=torch.zeros(b, dtype=torch.int64, device=DEV))
    empty_b = RaggedBatch(torch.zeros(b, 0, 4, device=DEV), sample_sizes=torch.zeros(b, dtype=torch.int64, device=DEV))
    hm = [torch.full((b, 64, 128), 0.5, device=DEV)]
    ops.draw_heatmap_multiscale(hm, empty_c, empty_b, (4.0,), clear=False, _sampler_job=job)
    assert "splat_multi_sampler_kernel" in nat.last_dispatch() and "grid(6,1,1)" in nat.last_dispatch()
    assert (hm[0] == 0.5).all()
    assert _same_samples(job.samples, sample_lanes(pts, q))
    # odd map width: the box maps fall back to the per-scale operators and the job runs as the stand-alone sampler
    centers, boxes = _objects(b, 5, 500.0, 300.0, seed=2)
    job2 = lanes._SamplerJob(pts, None, q, work)
    ops.draw_heatmap_multiscale([torch.zeros(b, 75, 125, device=DEV)], centers, boxes, (4.0,), clear=True, _sampler_job=job2)
    assert _same_samples(job2.samples, sample_lanes(pts, q))
    # C entry point: shapes the wave-level sampler does not take
    lib = nat.lib()
    ptrs = (ctypes.c_void_p * 1)(hm[0].data_ptr())
    hs, ws, st = (ctypes.c_int * 1)(64), (ctypes.c_int * 1)(128), (ctypes.c_float * 1)(4.0)
    cnt = torch.zeros(b, dtype=torch.int32, device=DEV)
    out = torch.zeros(b * l * q * 2 + 64, device=DEV)

    def call(points, samples_n, samples_ptr=out.data_ptr()):
        return lib.accv_draw_heatmap_multiscale_sample_f32(ptrs, hs, ws, st, 1, b, None, None, cnt.data_ptr(), 0, 6.0, 1.0, 0,
                                                           pts.data_ptr(), b * l, points, None, samples_n, samples_ptr,
                                                           work.data_ptr(), nat.stream_ptr(DEV))
    assert call(65, 64) == -1 and b"64 points" in lib.accv_last_error()
    assert call(9, 100) == -1 and b"multiple of 64" in lib.accv_last_error()
    assert call(9, 64, None) == -1 and b"null" in lib.accv_last_error()
    assert call(9, 64) == 0


def test_targets_replay_in_a_graph():
    from accvlab.draw_heatmap import draw_targets_multiscale

    b, sw, sh = 2, 1024.0, 512.0
    strides = (4.0, 8.0)
    centers, boxes = _objects(b, 10, sw, sh, seed=5)
    pts, _, _ = _polylines(b, 6, 20, sw, sh, seed=6, ragged=False)
    shapes = [(b, int(sh / s), int(sw / s)) for s in strides]
    box_r, lane_r = [torch.zeros(s_, device=DEV) for s_ in shapes], [torch.zeros(s_, device=DEV) for s_ in shapes]
    draw_targets_multiscale(box_r, centers, boxes, strides, lane_r, pts, 128, 2, clear=True)      # (fills the constant cache)
    torch.cuda.synchronize()
    box_g, lane_g = [torch.zeros(s_, device=DEV) for s_ in shapes], [torch.zeros(s_, device=DEV) for s_ in shapes]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            draw_targets_multiscale(box_g, centers, boxes, strides, lane_g, pts, 128, 2, clear=True)
    for m in box_g + lane_g:
        m.fill_(3.0)
    graph.replay()
    torch.cuda.synchronize()
    for a, r in zip(box_g + lane_g, box_r + lane_r):
        assert torch.equal(a, r)
