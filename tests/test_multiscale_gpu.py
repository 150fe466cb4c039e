"""draw_heatmap_multiscale (extension; BASELINE config 3): all strides of a batch in one launch == the per-scale
composition get_centers_and_radii + draw_heatmap_batched (bit for bit: same arithmetic in the same order per pixel),
and == the CPU oracle fed with the reference front-end rule (packages/draw_heatmap/tests/_test_helpers.py:20-28)."""
import numpy as np
import pytest
import torch

from oracle import h1 as oracle

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _objects(b, n_max, sw, sh, seed):
    from accvlab.batching_helpers import combine_data

    g = torch.Generator().manual_seed(seed)
    cs, bs = [], []
    for _ in range(b):
        n = int(torch.randint(0, n_max + 1, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([sw, sh])
        half = torch.rand(n, 4, generator=g) * min(sw, sh) * 0.15
        cs.append(c)
        bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
    crb = combine_data(cs, device=DEV)
    brb = combine_data(bs, device=DEV, other_with_same_sample_sizes=crb)
    return crb, brb


@pytest.mark.parametrize("clear", [True, False])
@pytest.mark.parametrize("strides,sw,sh", [((4.0, 8.0, 16.0), 960, 544), ((2.0,), 256, 64), ((1.0, 3.0, 4.0, 32.0), 384, 96)])
def test_multiscale_equals_per_scale_calls_and_oracle(clear, strides, sw, sh):
    from accvlab.draw_heatmap import draw_heatmap_batched, draw_heatmap_multiscale, get_centers_and_radii

    b = 5
    crb, brb = _objects(b, 24, sw, sh, seed=int(sum(strides)))
    shapes = [(b, -(-int(sh / s) // 1), -(-int(sw / s) // 4) * 4) for s in strides]     # widths: multiples of 4
    base = [torch.rand(sh_, generator=torch.Generator().manual_seed(i)).mul_(0.2).to(DEV) for i, sh_ in enumerate(shapes)]
    fused = [t.clone() for t in base]
    draw_heatmap_multiscale(fused, crb, brb, strides, 6.0, 0.8, clear=clear)
    for i, s in enumerate(strides):
        ci, ri = get_centers_and_radii(crb, brb, s)
        ref = base[i].clone()
        draw_heatmap_batched(ref, ci, ri, 6.0, 0.8, clear=clear)
        assert torch.equal(fused[i], ref), f"stride {s}: fused result differs from the per-scale operators"
        want = base[i].cpu().numpy().copy()
        oracle.draw_heatmap_batched(want, ci.tensor.cpu().numpy(), ri.tensor.cpu().numpy(),
                                    crb.sample_sizes.cpu().numpy(), k=0.8, clear=clear)
        assert np.abs(fused[i].cpu().numpy() - want).max() <= 1e-5


def test_multiscale_fallback_shapes_and_validation():
    from accvlab.draw_heatmap import draw_heatmap_batched, draw_heatmap_multiscale, get_centers_and_radii

    b = 3
    crb, brb = _objects(b, 10, 300, 200, seed=9)
    # a width that is not a multiple of 4 and five scales: python falls back to the per-scale operators
    strides = (2.0, 3.0, 4.0, 5.0, 7.0)
    maps = [torch.zeros(b, int(200 / s), int(300 / s), device=DEV) for s in strides]
    draw_heatmap_multiscale(maps, crb, brb, strides, clear=True)
    for hm, s in zip(maps, strides):
        ci, ri = get_centers_and_radii(crb, brb, s)
        ref = torch.empty_like(hm)
        draw_heatmap_batched(ref, ci, ri, clear=True)
        assert torch.equal(hm, ref)
    with pytest.raises(RuntimeError):
        draw_heatmap_multiscale(maps[:2], crb, brb, (2.0,))                         # length mismatch
    with pytest.raises(RuntimeError):
        draw_heatmap_multiscale([maps[0].double()], crb, brb, (2.0,))               # dtype
    with pytest.raises(RuntimeError):
        draw_heatmap_multiscale([maps[0][:2]], crb, brb, (2.0,))                    # batch mismatch
    # no objects at all + clear -> zeros
    from accvlab.batching_helpers import combine_data
    empty_c = combine_data([torch.zeros(0, 2)] * b, device=DEV)
    empty_b = combine_data([torch.zeros(0, 4)] * b, device=DEV)
    hm = torch.full((b, 16, 32), 3.0, device=DEV)
    draw_heatmap_multiscale([hm], empty_c, empty_b, (4.0,), clear=True)
    assert (hm == 0).all()


def test_multiscale_is_graph_capturable():
    from accvlab.draw_heatmap import draw_heatmap_multiscale

    crb, brb = _objects(4, 16, 640, 384, seed=3)
    maps = [torch.zeros(4, 96, 160, device=DEV), torch.zeros(4, 48, 80, device=DEV)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        draw_heatmap_multiscale(maps, crb, brb, (4.0, 8.0), clear=True)
    torch.cuda.synchronize()
    want = [m.clone() for m in maps]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        draw_heatmap_multiscale(maps, crb, brb, (4.0, 8.0), clear=True)
    for m in maps:
        m.fill_(7.0)
    g.replay()
    torch.cuda.synchronize()
    for m, w in zip(maps, want):
        assert torch.equal(m, w)


def test_extension_module_entry_points_match_the_public_operators():
    # accvlab.draw_heatmap.draw_heatmap_ext: the names of the reference's compiled module (csrc/draw_heatmap.cpp:131-144)
    from types import SimpleNamespace

    from accvlab.draw_heatmap import draw_heatmap_batched
    from accvlab.draw_heatmap import draw_heatmap_ext as ext

    g = torch.Generator().manual_seed(2)
    b, n, c, h, w = 3, 7, 4, 40, 64
    centers = torch.stack([torch.randint(0, w, (b, n), generator=g), torch.randint(0, h, (b, n), generator=g)], -1)
    centers = centers.to(torch.int32).to(DEV)
    radii = torch.randint(1, 9, (b, n), generator=g).to(torch.int32).to(DEV)
    labels = torch.randint(0, c, (b, n), generator=g).to(torch.int32).to(DEV)
    counts = torch.tensor([7, 0, 3], dtype=torch.int32, device=DEV)
    rb = lambda t: SimpleNamespace(tensor=t, sample_sizes=counts)  # noqa: E731
    a, ref = torch.zeros(b, h, w, device=DEV), torch.zeros(b, h, w, device=DEV)
    ext.draw_heatmap_batched_impl(a, centers, radii, counts, 6.0, 0.7)
    draw_heatmap_batched(ref, rb(centers), rb(radii), 6.0, 0.7)
    assert torch.equal(a, ref) and float(a.max()) > 0
    a, ref = torch.zeros(b, c, h, w, device=DEV), torch.zeros(b, c, h, w, device=DEV)
    ext.draw_heatmap_batched_classwise_impl(a, centers, radii, counts, labels, 6.0, 0.7)
    draw_heatmap_batched(ref, rb(centers), rb(radii), 6.0, 0.7, rb(labels))
    assert torch.equal(a, ref) and float(a.max()) > 0


@pytest.mark.parametrize("clear", [True, False])
def test_store_policy_hints_do_not_change_results(clear):
    # write_through (ACCV_HM_WRITE_THROUGH) and small_radii (ACCV_HM_SMALL_RADII) select other kernels / store
    # instructions, never other values
    from types import SimpleNamespace

    from accvlab.draw_heatmap import draw_heatmap, draw_heatmap_batched

    g = torch.Generator().manual_seed(4)
    b, n, h, w = 4, 11, 70, 132
    centers = torch.stack([torch.randint(-3, w + 3, (b, n), generator=g), torch.randint(-3, h + 3, (b, n), generator=g)], -1)
    centers = centers.to(torch.int32).to(DEV)
    radii = torch.randint(0, 30, (b, n), generator=g).to(torch.int32).to(DEV)
    counts = torch.tensor([11, 0, 5, 9], device=DEV)
    rb = lambda t: SimpleNamespace(tensor=t, sample_sizes=counts)  # noqa: E731
    base = torch.rand(b, h, w, generator=g).mul_(0.2).to(DEV)
    ref = base.clone()
    draw_heatmap_batched(ref, rb(centers), rb(radii), clear=clear)
    for kw in ({"write_through": True}, {"write_through": False}, {"small_radii": True},
               {"write_through": True, "small_radii": True}):
        got = base.clone()
        draw_heatmap_batched(got, rb(centers), rb(radii), clear=clear, **kw)
        if kw.get("small_radii"):
            assert float((got - ref).abs().max()) <= 1e-6       # exp of the sum vs product of two exps
        else:
            assert torch.equal(got, ref)
    idx = torch.arange(b, dtype=torch.int32, device=DEV).repeat_interleave(n)
    flat_ref, flat_wt = base.clone(), base.clone()
    draw_heatmap(flat_ref, centers.reshape(-1, 2).contiguous(), radii.reshape(-1).contiguous(), idx, clear=clear)
    draw_heatmap(flat_wt, centers.reshape(-1, 2).contiguous(), radii.reshape(-1).contiguous(), idx, clear=clear,
                 write_through=True)
    assert torch.equal(flat_ref, flat_wt)


def test_density_adaptive_store_policy_is_value_neutral():
    """in-place launches pick write-through or plain stores PER PLANE from sum (2r+1)^2 of the plane's objects (>= 3/4 of
    the plane's area -> write-through): a batch that mixes dense and sparse planes must equal the forced-plain and the
    forced-write-through results bit for bit, class-wise too"""
    from types import SimpleNamespace

    from accvlab.draw_heatmap import draw_heatmap_batched

    g = torch.Generator().manual_seed(11)
    b, n, h, w = 6, 40, 96, 256
    centers = torch.stack([torch.randint(0, w, (b, n), generator=g), torch.randint(0, h, (b, n), generator=g)], -1).to(torch.int32)
    radii = torch.randint(1, 4, (b, n), generator=g).to(torch.int32)
    radii[0] = 60                                       # plane 0: 40 objects of 121 x 121 -> far above the threshold
    radii[3, :2] = 80                                   # plane 3: two large objects -> 2 * 161^2 = 2.1 x the plane
    counts = torch.tensor([40, 40, 0, 2, 7, 40])
    labels = torch.randint(0, 3, (b, n), generator=g).to(torch.int32)
    rb = lambda t: SimpleNamespace(tensor=t.to(DEV), sample_sizes=counts.to(DEV))  # noqa: E731
    base = torch.rand(b, h, w, generator=g).mul_(0.3).to(DEV)
    outs = []
    for wt in (None, False, True):
        hm = base.clone()
        draw_heatmap_batched(hm, rb(centers), rb(radii), 6.0, 0.9, write_through=wt)
        outs.append(hm)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert bool((outs[0] != base).any())
    base4 = torch.rand(b, 3, h, w, generator=g).mul_(0.3).to(DEV)
    outs = []
    for wt in (None, False, True):
        hm = base4.clone()
        draw_heatmap_batched(hm, rb(centers), rb(radii), 6.0, 0.9, rb(labels), write_through=wt)
        outs.append(hm)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("clear", [True, False])
def test_scale_order_inside_the_launch_does_not_change_results(clear):
    """Default: coarse scales are dispatched first; ACCV_HM_CALLER_SCALE_ORDER keeps the caller's order.  Same maps."""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import draw_heatmap_multiscale, draw_polylines_multiscale, ops

    b, sw, sh = 4, 960, 544
    strides = (4.0, 16.0, 8.0)     # deliberately unsorted
    crb, brb = _objects(b, 24, sw, sh, seed=21)
    base = [torch.rand(b, int(sh / s), int(sw / s), generator=torch.Generator().manual_seed(i)).mul_(0.2).to(DEV)
            for i, s in enumerate(strides)]
    g = torch.Generator().manual_seed(5)
    lanes = (torch.rand(b, 3, 9, 2, generator=g) * torch.tensor([sw, sh])).to(DEV)
    results = {}
    for label, flag in (("coarse first", 0), ("caller order", nat.HM_CALLER_SCALE_ORDER)):
        ops._FORCED_FLAGS = flag
        try:
            boxes = [t.clone() for t in base]
            draw_heatmap_multiscale(boxes, crb, brb, strides, 6.0, 0.8, clear=clear)
            assert "splat_multi_kernel" in nat.last_dispatch()
            lane_maps = [t.clone() for t in base]
            draw_polylines_multiscale(lane_maps, lanes, 64, 2, strides, clear=clear)
            assert "lane_raster_multi_kernel" in nat.last_dispatch()      # 3 polylines x 9 points: the fused lane raster
            more = [t.clone() for t in base]                              # ... and sampler + point splat (5 x 9 points)
            draw_polylines_multiscale(more, torch.cat([lanes, lanes[:, :2] * 0.5], 1), 64, 2, strides, clear=clear)
            assert "splat_points_multi_kernel" in nat.last_dispatch()
            lane_maps += more
        finally:
            ops._FORCED_FLAGS = 0
        results[label] = boxes + lane_maps
    for a, c in zip(results["coarse first"], results["caller order"]):
        assert torch.equal(a, c)
