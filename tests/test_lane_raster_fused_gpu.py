"""Fused lane raster (`lane_raster_multi_kernel`: the tile waves sample the polylines themselves) — GPU parity.

The fused launch must write BIT FOR BIT what the two-launch composition writes (polyline sampler -> point splat), which in turn
is pinned stage by stage against the lane / heat-map oracles in tests/test_lane_raster_gpu.py.  Covered here: shapes on both
sides of every limit of the fused kernel (points per polyline 1..64, 64 point slots per frame, samples 1..1000), ragged point and
lane counts, clear and in-place, radii 0..9, and the inputs the segment cull must not mishandle (NaN / inf / huge coordinates,
zero-length polylines and segments, single points, polylines that re-enter a tile, polylines far outside the maps)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _polylines(b, l, p, w, h, seed, ragged, wiggle=1.0):
    g = np.random.default_rng(seed)
    start = g.uniform([0, 0], [w, h], size=(b, l, 1, 2))
    steps = g.normal(0, wiggle, size=(b, l, p, 2)) * [w / p / 2, h / p / 2] + [w / p / 3, -h / p / 4]
    pts = (start + np.cumsum(steps, axis=2)).astype(np.float32)
    npts = g.integers(0, p + 1, size=(b, l)).astype(np.int64) if ragged else None
    nlanes = g.integers(0, l + 1, size=(b,)).astype(np.int32) if ragged else None
    return pts, npts, nlanes


def _both_ways(maps_shape, strides, pts, q, radius, npts=None, nlanes=None, clear=True, expect_fused=True, seed=0):
    """draw with the fused kernel and with sampler + point splat; returns the fused maps after asserting equality"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_polylines_multiscale, lanes

    dev = torch.device("cuda", 0)
    pts_d = torch.from_numpy(pts).to(dev) if isinstance(pts, np.ndarray) else pts
    npts_d = torch.from_numpy(npts).to(dev) if npts is not None else None
    nlanes_d = torch.from_numpy(nlanes).to(dev) if nlanes is not None else None
    base = [torch.rand(s_, generator=torch.Generator().manual_seed(seed + i)).mul_(0.3).to(dev) for i, s_ in enumerate(maps_shape)]
    one, two = [t.clone() for t in base], [t.clone() for t in base]
    assert lanes.FUSED_SAMPLER
    if expect_fused is None:
        k = len(maps_shape)
        hs, ws = (ctypes.c_int * k)(*[s_[1] for s_ in maps_shape]), (ctypes.c_int * k)(*[s_[2] for s_ in maps_shape])
        expect_fused = bool(nat.lib().accv_draw_polylines_fused_applicable(hs, ws, k, pts_d.shape[0], pts_d.shape[1],
                                                                          pts_d.shape[2], q))
    draw_polylines_multiscale(one, pts_d, q, radius, strides, 6.0, 0.9, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
    name = nat.last_dispatch()
    assert ("lane_raster_multi_kernel" in name) == expect_fused, name
    _both_ways.fused_seen = getattr(_both_ways, "fused_seen", 0) + int(expect_fused)
    lanes.FUSED_SAMPLER = False
    try:
        draw_polylines_multiscale(two, pts_d, q, radius, strides, 6.0, 0.9, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
        assert "splat_points_multi_kernel" in nat.last_dispatch()
    finally:
        lanes.FUSED_SAMPLER = True
    for i, (a, b_) in enumerate(zip(one, two)):
        if not torch.equal(a, b_):
            bad = (a != b_).nonzero()
            raise AssertionError(f"scale {i} (stride {strides[i]}): {bad.shape[0]} pixels differ, first {bad[0].tolist()}: "
                                 f"{a[tuple(bad[0])].item()} vs {b_[tuple(bad[0])].item()}")
    return one, base


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("clear", [True, False])
@pytest.mark.parametrize("b,l,p", [(2, 2, 24), (3, 1, 64), (2, 16, 4), (2, 4, 16), (2, 5, 2), (2, 1, 33), (2, 7, 3), (2, 8, 8),
                                   (2, 8, 24)])
@pytest.mark.parametrize("q,radius", [(256, 2), (17, 0), (64, 5), (1000, 1), (1, 3), (2, 9)])
def test_fused_equals_sampler_plus_point_splat(ragged, clear, b, l, p, q, radius):
    sw, sh = 3072.0, 1728.0
    strides = (4.0, 8.0, 16.0)
    pts, npts, nlanes = _polylines(b, l, p, sw, sh, seed=b * 1000 + l * 10 + p + q, ragged=ragged)
    shapes = [(b, int(sh / s), int(sw / s)) for s in strides]
    # (expect_fused=None asks the library which kernel applies: the fused one takes frames of at most 64 point slots whose
    # segments carry few samples each; the sweep must have exercised it — checked by the test below)
    one, base = _both_ways(shapes, strides, pts, q, radius, npts, nlanes, clear, expect_fused=None)
    if not ragged:
        assert any(bool((f != b_).any()) for f, b_ in zip(one, base))            # something was drawn


def test_sweep_exercised_the_fused_kernel():
    assert getattr(_both_ways, "fused_seen", 0) >= 100, "the parameter sweep above no longer reaches the fused kernel"


def test_fused_kernel_shape_limits():
    """shapes outside the fused kernel take the two-launch path (and the C entry point refuses them)"""
    from accvlab import _amd_native as nat

    sw, sh = 1024.0, 512.0
    shapes, strides = [(2, 512, 1024)], (1.0,)
    # 64 point slots per frame (points rounded up to a power of two, at least 4), at most (points - 1) x slots / 2 samples
    for l, p, q, fused in ((1, 64, 32, True), (2, 64, 32, False), (1, 65, 32, False), (16, 4, 6, True), (17, 4, 6, False),
                           (16, 4, 7, False), (4, 16, 32, True), (5, 16, 32, False), (2, 17, 32, True), (3, 17, 32, False),
                           (8, 8, 28, True), (8, 8, 29, False), (9, 8, 28, False), (2, 24, 368, True), (2, 24, 369, False),
                           (3, 1, 2, True), (3, 1, 3, False)):
        pts, _, _ = _polylines(2, l, p, sw, sh, seed=l * 100 + p, ragged=False)
        _both_ways(shapes, strides, pts, q, 2, expect_fused=fused)
    lib = nat.lib()
    hs, ws = (ctypes.c_int * 1)(512), (ctypes.c_int * 1)(1024)
    assert lib.accv_draw_polylines_fused_applicable(hs, ws, 1, 2, 1, 64, 96) == 1
    assert lib.accv_draw_polylines_fused_applicable(hs, ws, 1, 2, 2, 64, 96) == 0
    assert lib.accv_draw_polylines_fused_applicable(hs, ws, 1, 2, 1, 64, 63 * 32) == 1
    assert lib.accv_draw_polylines_fused_applicable(hs, ws, 1, 2, 1, 64, 63 * 32 + 1) == 0
    # coarse scales in the majority of the tiles: the point splat's four-waves-per-tile shape is kept
    hs2, ws2 = (ctypes.c_int * 1)(16), (ctypes.c_int * 1)(128)
    assert lib.accv_draw_polylines_fused_applicable(hs2, ws2, 1, 2, 2, 24, 256) == 0
    hs3, ws3 = (ctypes.c_int * 1)(256), (ctypes.c_int * 1)(1024)
    assert lib.accv_draw_polylines_fused_applicable(hs3, ws3, 1, 2, 2, 24, 256) == 1
    dev = torch.device("cuda", 0)
    hm = torch.zeros(2, 512, 1024, device=dev)
    pts = torch.zeros(2, 5, 64, 2, device=dev)
    lanes_n = torch.full((2,), 1, dtype=torch.int32, device=dev)
    ptrs = (ctypes.c_void_p * 1)(hm.data_ptr())
    st = (ctypes.c_float * 1)(1.0)
    rc = lib.accv_draw_polylines_multiscale_f32(ptrs, hs, ws, st, 1, 2, pts.data_ptr(), 5, 64, None, lanes_n.data_ptr(),
                                                96, 2, 6.0, 1.0, nat.HM_CLEAR, nat.stream_ptr(dev))
    assert rc == -1 and b"fused" in lib.accv_last_error()
    rc = lib.accv_draw_polylines_multiscale_f32(ptrs, hs, ws, st, 1, 2, pts.data_ptr(), 1, 64, None, None,
                                                96, 2, 6.0, 1.0, nat.HM_CLEAR, nat.stream_ptr(dev))
    assert rc == -1 and b"null" in lib.accv_last_error()
    rc = lib.accv_draw_polylines_multiscale_f32(ptrs, hs, ws, st, 1, 2, pts.data_ptr(), 1, 64, None, lanes_n.data_ptr(),
                                                96, -1, 6.0, 1.0, nat.HM_CLEAR, nat.stream_ptr(dev))
    assert rc == -1 and b"radius" in lib.accv_last_error()


@pytest.mark.parametrize("clear", [True, False])
def test_fused_segment_cull_edge_inputs(clear):
    """inputs the segment-level cull must not mishandle"""
    sw, sh = 1024.0, 512.0
    strides = (2.0, 8.0)
    nb, nl, npnt = 5, 4, 16
    shapes = [(nb, int(sh / s), int(sw / s)) for s in strides]
    pts, _, _ = _polylines(nb, nl, npnt, sw, sh, seed=77, ragged=False)
    nan, inf = float("nan"), float("inf")
    t = np.arange(npnt)
    # frame 0: NaN / inf / huge coordinates inside otherwise ordinary polylines
    pts[0, 0, 7, 0] = nan
    pts[0, 1, 0, 1] = inf
    pts[0, 2, npnt - 1, 0] = -inf
    pts[0, 3, 5] = (3.0e6, 2.0e6)
    # frame 1: more of the same, a zero-length polyline, repeated points (zero-length segments)
    pts[1, 0, 11] = (-1.5e9, 40.0)
    pts[1, 1, :] = (300.25, 200.75)
    pts[1, 2, 5:12] = pts[1, 2, 5]
    pts[1, 3, 3] = (1.0e30, -1.0e30)
    # frame 2: a polyline far outside, polylines on the map borders
    pts[2, 0] += (50000.0, -30000.0)
    pts[2, 1, :, 1] = 0.0
    pts[2, 2, :, 0] = sw - 0.5
    pts[2, 3, :, 0] = -0.25
    # frame 3: zig-zags that leave and re-enter the same tiles, tiny segments next to long ones
    pts[3, 0, :, 0] = 100.0 + 400.0 * (t % 2)
    pts[3, 0, :, 1] = 100.0 + 3.0 * t
    pts[3, 1, :, 0] = 500.0 + 1e-3 * t
    pts[3, 1, :, 1] = 250.0 + 1e-3 * (t % 3)
    pts[3, 2, :10] = pts[3, 2, 0] + 1e-6 * np.arange(10)[:, None]
    pts[3, 3, :, 0] = np.where(t < 8, 10.0 + t, 900.0 - t)
    pts[3, 3, :, 1] = 256.0
    # frame 4: ordinary polylines of 1, 0, 2 and all points
    npts = np.full((nb, nl), npnt, np.int64)
    npts[4, 0], npts[4, 1], npts[4, 2] = 1, 0, 2
    for q, radius in ((100, 2), (33, 4), (1, 1)):
        _both_ways(shapes, strides, pts, q, radius, npts=npts, clear=clear)


def test_fused_sparse_scene_at_config3_size_and_count_dtypes():
    """maps of BASELINE configs[3] (strides 4 / 8 / 16 of a 3840 x 2160 source) with two polylines of 24 points per frame, 256
    samples, radius 2; int32 / int64 counts take the same kernel"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_polylines_multiscale

    b, sw, sh = 4, 3840.0, 2160.0
    strides = (4.0, 8.0, 16.0)
    shapes = [(b, int(sh / s), int(sw / s)) for s in strides]
    pts, _, _ = _polylines(b, 2, 24, sw, sh, seed=3, ragged=False, wiggle=0.3)
    one, base = _both_ways(shapes, strides, pts, 256, 2)
    # int32 point counts / int64 lane counts take the same kernel
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(5)
    npts = torch.from_numpy(g.integers(0, 25, size=(b, 2)).astype(np.int32)).to(dev)
    nl = torch.from_numpy(g.integers(0, 3, size=(b,)).astype(np.int64)).to(dev)
    pts_d = torch.from_numpy(pts).to(dev)
    a = [torch.empty(s_, device=dev) for s_ in shapes]
    draw_polylines_multiscale(a, pts_d, 256, 2, strides, num_points=npts, num_lanes=nl, clear=True)
    assert "lane_raster_multi_kernel" in nat.last_dispatch()
    c = [torch.empty(s_, device=dev) for s_ in shapes]
    draw_polylines_multiscale(c, pts_d, 256, 2, strides, num_points=npts.long(), num_lanes=nl.int(), clear=True)
    for x, y in zip(a, c):
        assert torch.equal(x, y)


def test_fused_lane_raster_replays_in_a_graph():
    from accvlab.draw_heatmap import draw_polylines_multiscale

    dev = torch.device("cuda", 0)
    strides = (4.0, 8.0)
    pts, _, _ = _polylines(2, 4, 12, 1024.0, 512.0, seed=9, ragged=False)
    pts_d = torch.from_numpy(pts).to(dev)
    maps = [torch.zeros(2, int(512 / s), int(1024 / s), device=dev) for s in strides]
    ref = [m.clone() for m in maps]
    from accvlab import _amd_native as nat
    draw_polylines_multiscale(ref, pts_d, 64, 2, strides, clear=True)      # also fills the constant cache outside the capture
    assert "lane_raster_multi_kernel" in nat.last_dispatch()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            draw_polylines_multiscale(maps, pts_d, 64, 2, strides, clear=True)
    for m in maps:
        m.fill_(7.0)
    graph.replay()
    torch.cuda.synchronize()
    for m, r in zip(maps, ref):
        assert torch.equal(m, r)
