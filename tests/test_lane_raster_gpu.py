"""Lane raster (BASELINE config 3: "lane_helpers polyline raster") — GPU parity.

The reference has no polyline rasteriser, so there is no reference output to pin the composed op against; parity is
checked stage by stage against the two pinned oracles it is built from:
  1. the arc-length samples                      vs  oracle.lane.sample (pinned by tests/golden/lane_polyline.npz), 1e-5 (scaled)
  2. float sample -> int centre / radius         vs  the rule of _test_helpers.py:20-28 (trunc(p / stride)), bit-exact
  3. the drawn map from THOSE integer targets    vs  oracle.h1 (pinned by tests/golden/h1_g*.npz), 1e-5 abs
"""
import numpy as np
import pytest
import torch

from oracle import h1 as oracle_h1
from oracle import lane as oracle_lane

pytestmark = pytest.mark.gpu


def _lanes(b, l, p, w, h, seed, ragged):
    g = np.random.default_rng(seed)
    start = g.uniform([0, 0], [w, h], size=(b, l, 1, 2))
    steps = g.normal(0, 1, size=(b, l, p, 2)) * [w / p / 2, h / p / 2] + [w / p / 3, -h / p / 4]
    pts = (start + np.cumsum(steps, axis=2)).astype(np.float32)
    npts = g.integers(0, p + 1, size=(b, l)).astype(np.int64) if ragged else None
    nlanes = g.integers(0, l + 1, size=(b,)).astype(np.int64) if ragged else None
    return pts, npts, nlanes


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("stride,radius,q", [(4.0, 2, 64), (1.0, 1, 17), (16.0, 3, 1)])
def test_lane_raster_stages(ragged, stride, radius, q):
    from accvlab.draw_heatmap import draw_polylines_batched, sample_lane_targets
    from accvlab.lane_helpers.polyline import interpolate_var_size_batch
    from accvlab.batching_helpers import RaggedBatch

    dev = torch.device("cuda", 0)
    b, l, p, sw, sh = 3, 5, 12, 960.0, 540.0
    h, w = int(sh / stride), int(sw / stride)
    pts, npts, nlanes = _lanes(b, l, p, sw, sh, seed=int(stride) * 10 + q, ragged=ragged)
    pts_d = torch.from_numpy(pts).to(dev)
    npts_d = torch.from_numpy(npts).to(dev) if ragged else None
    nlanes_d = torch.from_numpy(nlanes).to(dev) if ragged else None

    # stage 1: samples vs the lane oracle
    frac = np.linspace(0.0, 1.0, q).astype(np.float32) if q > 1 else np.zeros(1, np.float32)
    pr = RaggedBatch(pts_d.view(b * l, p, 2), sample_sizes=(npts_d.view(-1) if ragged else
                                                            torch.full((b * l,), p, device=dev, dtype=torch.int64)))
    dr = RaggedBatch(torch.from_numpy(np.tile(frac, (b * l, 1))).to(dev),
                     sample_sizes=torch.full((b * l,), q, device=dev, dtype=torch.int64))
    got = interpolate_var_size_batch(pr, dr, relative=True).tensor.cpu().numpy().reshape(b, l, q, 2)
    for i in range(b):
        for j in range(l):
            n = int(npts[i, j]) if ragged else p
            want = oracle_lane.sample(pts[i, j, :n], frac, relative=True)
            if n == 0:
                assert np.isnan(got[i, j]).all()
            else:
                assert np.abs(got[i, j] - want).max() <= 1e-5 * max(sw, sh)   # coordinates are O(1e3): relative 1e-5

    # stage 2: integer targets from the kernel's own samples, bit-exact
    centers, radii = sample_lane_targets(pts_d, q, radius, stride, num_points=npts_d)
    c = centers.cpu().numpy().reshape(b, l, q, 2)
    r = radii.cpu().numpy().reshape(b, l, q)
    bad = np.isnan(got).any(-1)
    want_c = np.where(bad[..., None], 0, np.trunc(np.nan_to_num(got) / np.float32(stride))).astype(np.int32)
    assert np.array_equal(c, want_c)
    assert np.array_equal(r, np.where(bad, -1, radius))

    # stage 3: composed draw vs the heat-map oracle on the same integer targets (in place on a non-zero map, and clear)
    sizes = (nlanes * q) if ragged else np.full(b, l * q, dtype=np.int64)
    base = np.random.default_rng(1).uniform(0, 0.3, size=(b, h, w)).astype(np.float32)
    for clear in (False, True):
        hm = torch.from_numpy(base.copy()).to(dev)
        draw_polylines_batched(hm, pts_d, q, radius, stride, 6.0, 0.9, num_points=npts_d, num_lanes=nlanes_d,
                               clear=clear)
        ref = base.copy()
        oracle_h1.draw_heatmap_batched(ref, centers.cpu().numpy(), radii.cpu().numpy(), sizes, k=0.9, clear=clear)
        assert np.abs(hm.cpu().numpy() - ref).max() <= 1e-5
        if not ragged and q > 1:
            assert (hm.cpu().numpy() > base if not clear else hm.cpu().numpy() > 0).any()   # something was drawn


def test_lane_raster_validation_and_empty():
    from accvlab.draw_heatmap import draw_polylines_batched, sample_lane_targets

    dev = torch.device("cuda", 0)
    with pytest.raises(RuntimeError):
        sample_lane_targets(torch.zeros(2, 3, 4, 2), 8, 1)                       # CPU tensor
    with pytest.raises(RuntimeError):
        sample_lane_targets(torch.zeros(2, 3, 4, 3, device=dev), 8, 1)           # not [.., 2]
    with pytest.raises(RuntimeError):
        sample_lane_targets(torch.zeros(2, 3, 4, 2, device=dev, dtype=torch.float64), 8, 1)
    with pytest.raises(RuntimeError):
        sample_lane_targets(torch.zeros(2, 3, 4, 2, device=dev), 0, 1)
    c, r = sample_lane_targets(torch.zeros(2, 0, 4, 2, device=dev), 8, 1)
    assert c.shape == (2, 0, 2) and r.shape == (2, 0)
    hm = torch.full((2, 16, 16), 0.25, device=dev)
    draw_polylines_batched(hm, torch.zeros(2, 0, 4, 2, device=dev), 8, 1)       # no lanes: map untouched
    assert (hm == 0.25).all()
    # all lanes empty (zero points) -> NaN samples -> nothing drawn
    draw_polylines_batched(hm, torch.zeros(2, 3, 4, 2, device=dev), 8, 2,
                           num_points=torch.zeros(2, 3, dtype=torch.int32, device=dev))
    assert (hm == 0.25).all()


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("clear", [True, False])
@pytest.mark.parametrize("radius,q", [(2, 64), (1, 200), (5, 17)])
def test_multiscale_lane_raster_equals_per_scale_calls(ragged, clear, radius, q):
    from accvlab.draw_heatmap import draw_polylines_batched, draw_polylines_multiscale
    from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets as via_targets

    dev = torch.device("cuda", 0)
    b, l, p, sw, sh = 3, 5, 12, 1536.0, 864.0
    strides = (4.0, 8.0, 16.0)
    pts, npts, nlanes = _lanes(b, l, p, sw, sh, seed=radius * 100 + q, ragged=ragged)
    pts_d = torch.from_numpy(pts).to(dev)
    npts_d = torch.from_numpy(npts).to(dev) if ragged else None
    nlanes_d = torch.from_numpy(nlanes).to(dev) if ragged else None
    shapes = [(b, int(sh / s), int(sw / s)) for s in strides]
    base = [torch.rand(s_, generator=torch.Generator().manual_seed(i)).mul_(0.3).to(dev) for i, s_ in enumerate(shapes)]
    fused = [t.clone() for t in base]
    draw_polylines_multiscale(fused, pts_d, q, radius, strides, 6.0, 0.9, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
    for i, s in enumerate(strides):
        ref = base[i].clone()
        via_targets(ref, pts_d, q, radius, s, 6.0, 0.9, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
        assert torch.equal(fused[i], ref), f"stride {s}: fused lane raster differs from the per-scale operator"
    assert any(bool((f != b_).any()) for f, b_ in zip(fused, base))            # something was drawn


def test_multiscale_lane_raster_fallback_and_large_radius():
    from accvlab.draw_heatmap import draw_polylines_batched, draw_polylines_multiscale
    from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets as via_targets

    dev = torch.device("cuda", 0)
    pts, _, _ = _lanes(2, 3, 9, 400.0, 300.0, seed=5, ragged=False)
    pts_d = torch.from_numpy(pts).to(dev)
    # odd widths -> python falls back to the per-scale operator (bit-identical by construction)
    maps = [torch.zeros(2, 75, 101, device=dev), torch.zeros(2, 37, 50, device=dev)]
    draw_polylines_multiscale(maps, pts_d, 40, 2, (4.0, 8.0), clear=True)
    for hm, s in zip(maps, (4.0, 8.0)):
        ref = torch.empty_like(hm)
        via_targets(ref, pts_d, 40, 2, s, clear=True)
        assert torch.equal(hm, ref)
    # a radius above the small-splat hint: the fused op keeps the box-walking arithmetic, the per-scale op switches to
    # the tile kernel (separable product) -> equal to rounding
    a = [torch.zeros(2, 76, 100, device=dev)]
    draw_polylines_multiscale(a, pts_d, 40, 12, (4.0,), clear=True)
    ref = torch.zeros(2, 76, 100, device=dev)
    via_targets(ref, pts_d, 40, 12, 4.0, clear=True)
    assert float((a[0] - ref).abs().max()) <= 1e-6


@pytest.mark.parametrize("q", [64, 192])
def test_sampler_group_boxes_match_the_samples(q):
    # accv_polyline_sample_boxes: bounding box of every 64 consecutive samples, NaN samples ignored, empty groups inverted
    from accvlab.draw_heatmap import sample_lanes

    dev = torch.device("cuda", 0)
    b, l, p = 3, 4, 10
    pts, npts, _ = _lanes(b, l, p, 800.0, 600.0, seed=q, ragged=True)
    npts[0, 1] = 0                                                   # one empty lane -> NaN samples -> empty groups
    pts_d, npts_d = torch.from_numpy(pts).to(dev), torch.from_numpy(npts).to(dev)
    groups = b * l * q // 64
    boxes = torch.full((groups, 4), 7.0, device=dev)
    samples = sample_lanes(pts_d, q, num_points=npts_d, group_boxes_ptr=boxes.data_ptr())
    torch.cuda.synchronize()
    ref = sample_lanes(pts_d, q, num_points=npts_d)
    assert torch.equal(torch.nan_to_num(samples, nan=-1.0), torch.nan_to_num(ref, nan=-1.0))
    s = samples.view(groups, 64, 2).cpu()
    got = boxes.cpu()
    inf = float("inf")
    for g in range(groups):
        ok = ~torch.isnan(s[g]).any(-1)
        if ok.any():
            v = s[g][ok]
            want = torch.tensor([v[:, 0].min(), v[:, 1].min(), v[:, 0].max(), v[:, 1].max()])
        else:
            want = torch.tensor([inf, inf, -inf, -inf])
        assert torch.equal(got[g], want), (g, got[g], want)


def test_lane_splat_one_and_four_waves_per_tile_agree():
    """splat_points_multi_kernel runs with one wave per tile when every scale is fine (few samples per tile) and with four
    waves sharing a tile when some scale is coarse; both must equal the per-scale operator bit for bit"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_polylines_batched, draw_polylines_multiscale
    from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets as via_targets

    dev = torch.device("cuda", 0)
    pts, npts, nlanes = _lanes(4, 6, 14, 2048.0, 1024.0, seed=21, ragged=True)
    pts_d, npts_d, nlanes_d = torch.from_numpy(pts).to(dev), torch.from_numpy(npts).to(dev), torch.from_numpy(nlanes).to(dev)
    seen = set()
    for strides in ((1.0,), (2.0, 4.0), (32.0,), (2.0, 64.0)):
        for clear in (True, False):
            shapes = [(4, int(1024 / s), int(2048 / s)) for s in strides]
            base = [torch.rand(s_, generator=torch.Generator().manual_seed(7)).mul_(0.2).to(dev) for s_ in shapes]
            fused = [b.clone() for b in base]
            draw_polylines_multiscale(fused, pts_d, 128, 2, strides, 6.0, 0.8, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
            seen.add("block(256)" in nat.last_dispatch())
            for i, s in enumerate(strides):
                ref = base[i].clone()
                via_targets(ref, pts_d, 128, 2, s, 6.0, 0.8, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
                assert torch.equal(fused[i], ref), (strides, s, clear)
    assert seen == {True, False}, "both launch shapes must have been exercised"


def test_single_scale_call_takes_the_point_splat_and_equals_the_three_launch_formulation():
    """draw_polylines_batched with a radius of a few pixels on an aligned map = draw_polylines_multiscale with one scale (the
    fused kernel, or the sampler + the point splat: two launches, two-level cull); other radii / alignments = sampler ->
    integer targets -> draw_heatmap_batched.  Bit-identical either way."""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_polylines_batched
    from accvlab.draw_heatmap.lanes import _draw_polylines_via_targets as via_targets

    dev = torch.device("cuda", 0)
    pts, npts, nlanes = _lanes(3, 5, 11, 1024.0, 512.0, seed=33, ragged=True)
    pts_d, npts_d, nlanes_d = torch.from_numpy(pts).to(dev), torch.from_numpy(npts).to(dev), torch.from_numpy(nlanes).to(dev)
    # (five polylines of 11 points are more than the 64 point slots of the fused kernel, tests/test_lane_raster_fused_gpu.py)
    for stride, radius, width, kernel in ((2.0, 2, 512, "splat_points_multi_kernel"), (4.0, 0, 256, "splat_points_multi_kernel"),
                                          (4.0, 12, 256, "splat_kernel"), (4.0, 2, 254, "splat_kernel")):
        for clear in (True, False):
            base = torch.rand(3, int(512 / stride), width, generator=torch.Generator().manual_seed(2)).mul_(0.3).to(dev)
            got, ref = base.clone(), base.clone()
            draw_polylines_batched(got, pts_d, 96, radius, stride, 6.0, 0.7, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
            assert kernel in nat.last_dispatch(), (stride, radius, width, nat.last_dispatch())
            via_targets(ref, pts_d, 96, radius, stride, 6.0, 0.7, num_points=npts_d, num_lanes=nlanes_d, clear=clear)
            assert torch.equal(got, ref), (stride, radius, width, clear)
