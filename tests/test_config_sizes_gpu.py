"""Every BASELINE.json config exercised at its STATED size on the GPU, against the oracle.

  configs[1]  draw_heatmap_batched, 64 x 1080 x 1920, ragged N in [1,128] via combine_data, rule A — the exact batch
              bench.py times — vs oracle/h1 (all host threads), fused-clear and in-place
  configs[2]  multi_tensor_copier, meta_tensor_tree(10_000) byte-exact host->GPU and GPU->host
  configs[3]  3840 x 2160 source, batch 32, strides 4/8/16: draw_heatmap_multiscale vs oracle/h1 on the integer targets
              of the reference front-end rule, and draw_polylines_multiscale vs oracle/lane + oracle/h1
  64-bit plane offsets: a class-wise map above 2^31 elements (64 x 20 x 1080 x 1920 = 2.65e9; the reference's `int`
              offsets overflow there, draw_heatmap_cuda_kernel.cuh:99-104), objects in the first and the last planes

Tolerance: 1e-5 abs for fp32 map values (north_star), bit-exact for integer targets and copied bytes.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import bench_workloads as wl
from oracle import h1 as oracle_h1
from oracle import lane as oracle_lane

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
ATOL = 1e-5


def _max_err(got: torch.Tensor, want: np.ndarray, chunk: int = 8) -> float:
    """max |got - want| without a second full-size host copy: compares `chunk` leading slices at a time"""
    err = 0.0
    for s in range(0, want.shape[0], chunk):
        g = got[s:s + chunk].cpu().numpy()
        err = max(err, float(np.abs(g - want[s:s + chunk]).max()))
    return err


# ------------------------------------------------------------------------------------------------ configs[1]
@pytest.mark.parametrize("clear", [True, False])
def test_c1_bench_batch_b64_vs_oracle(clear):
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_batched

    B, H, W = 64, 1080, 1920
    centers_l, radii_l = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=42)       # bench.py's rank-0 batch
    centers = combine_data([c.to(DEV) for c in centers_l])
    radii = combine_data([r.to(DEV) for r in radii_l])
    if clear:
        hm = torch.full((B, H, W), 123.0, device=DEV)                            # junk that must disappear
        want = np.zeros((B, H, W), dtype=np.float32)
    else:
        base = torch.rand((B, H, W), generator=torch.Generator().manual_seed(1)).mul_(0.5)
        want = base.numpy()
        hm = base.to(DEV)
    draw_heatmap_batched(hm, centers, radii, 6.0, 1.0, clear=clear)
    torch.cuda.synchronize()
    oracle_h1.draw_heatmap_batched(want, centers.tensor.cpu().numpy(), radii.tensor.cpu().numpy(),
                                   centers.sample_sizes.cpu().numpy(), clear=clear, threads=oracle_h1.max_threads())
    assert _max_err(hm, want) <= ATOL
    assert float(hm.max()) == 1.0                                                # every frame has >= 1 object centre


# ------------------------------------------------------------------------------------------------ configs[2]
def _leaves(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, dict):
        for v in x.values():
            yield from _leaves(v)
    elif isinstance(x, (list, tuple)):
        for v in x:
            yield from _leaves(v)


@pytest.mark.parametrize("background", [True, False])
def test_c2_ten_thousand_leaves_byte_exact_both_directions(background):
    from accvlab import multi_tensor_copier as mtc

    tree = wl.meta_tensor_tree(10_000, seed=0)
    src = list(_leaves(tree))
    assert len(src) == 10_000
    on_gpu = mtc.start_copy(tree, DEV, use_background_thread=background).get()
    got = list(_leaves(on_gpu))
    assert len(got) == len(src)
    for a, b in zip(src, got):
        assert b.device == DEV and a.dtype == b.dtype and a.shape == b.shape
    # byte-exact: one device-side comparison per leaf would be 10 k syncs; concatenate per dtype instead
    for dt in (torch.float32, torch.int64):
        a = torch.cat([t.reshape(-1) for t in src if t.dtype == dt])
        b = torch.cat([t.reshape(-1) for t in got if t.dtype == dt]).cpu()
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    assert on_gpu[0]["meta"]["name"] == "sample_0" and on_gpu[499]["meta"]["id"] == 499
    # and back: 10 000 small device tensors -> host
    back = mtc.start_copy(on_gpu, "cpu", use_background_thread=background).get()
    got_back = list(_leaves(back))
    assert len(got_back) == len(src)
    for a, b in zip(src, got_back):
        assert b.device.type == "cpu" and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)
    assert isinstance(back[3]["meta"]["aux"], tuple) and back[3]["meta"]["name"] == "sample_3"


# ------------------------------------------------------------------------------------------------ configs[3]
def _c3_boxes(b, n_max, sw, sh, seed):
    g = torch.Generator().manual_seed(seed)
    cs, bs = [], []
    for _ in range(b):
        n = int(torch.randint(1, n_max + 1, (1,), generator=g))
        c = torch.rand(n, 2, generator=g) * torch.tensor([sw, sh])
        half = torch.rand(n, 4, generator=g) * min(sw, sh) * 0.12 + 2.0
        cs.append(c)
        bs.append(torch.cat([c - half[:, :2], c + half[:, 2:]], 1))
    return cs, bs


def _front_end_rule(c: np.ndarray, box: np.ndarray, stride: float):
    """packages/draw_heatmap/tests/_test_helpers.py:20-28 in fp32: r = max(1, ceil(min edge distance / stride)),
    centre = int(c / stride)"""
    s = np.float32(stride)
    m = np.minimum(np.minimum(c[..., 0] - box[..., 0], c[..., 1] - box[..., 1]),
                   np.minimum(box[..., 2] - c[..., 0], box[..., 3] - c[..., 1])).astype(np.float32)
    r = np.maximum(1, np.ceil(m / s)).astype(np.int32)
    ci = np.trunc(c / s).astype(np.int32)
    return ci, r


@pytest.mark.parametrize("clear", [True, False])
def test_c3_multiscale_4k_batch32_vs_oracle(clear):
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_multiscale, get_centers_and_radii

    B, SW, SH = 32, 3840, 2160
    strides = (4.0, 8.0, 16.0)
    cs, bs = _c3_boxes(B, 64, SW, SH, seed=3)
    crb = combine_data(cs, device=DEV)
    brb = combine_data(bs, device=DEV, other_with_same_sample_sizes=crb)
    shapes = [(B, int(SH / s), int(SW / s)) for s in strides]
    assert shapes == [(32, 540, 960), (32, 270, 480), (32, 135, 240)]
    base = [torch.rand(s_, generator=torch.Generator().manual_seed(i)).mul_(0.3) for i, s_ in enumerate(shapes)]
    maps = [t.to(DEV) for t in base]
    draw_heatmap_multiscale(maps, crb, brb, strides, 6.0, 1.0, clear=clear)
    torch.cuda.synchronize()
    c_np, b_np = crb.tensor.cpu().numpy(), brb.tensor.cpu().numpy()
    sizes = crb.sample_sizes.cpu().numpy()
    for i, s in enumerate(strides):
        ci, ri = _front_end_rule(c_np, b_np, s)
        gi, gr = get_centers_and_radii(crb, brb, s)                              # the GPU front end: bit-exact
        valid = np.arange(c_np.shape[1])[None, :] < sizes[:, None]
        assert np.array_equal(gi.tensor.cpu().numpy()[valid], ci[valid])
        assert np.array_equal(gr.tensor.cpu().numpy()[valid], ri[valid])
        want = base[i].numpy().copy()
        oracle_h1.draw_heatmap_batched(want, ci, ri, sizes, clear=clear, threads=oracle_h1.max_threads())
        assert _max_err(maps[i], want) <= ATOL, f"stride {s}"
        assert float(maps[i].max()) >= 1.0 - 1e-6


@pytest.mark.parametrize("clear", [True, False])
def test_c3_lane_raster_4k_batch32_vs_oracles(clear):
    from accvlab.draw_heatmap import draw_polylines_multiscale

    B, L, P, SW, SH = 32, 8, 24, 3840.0, 2160.0
    strides, q_per_stride, radius = (4.0, 8.0, 16.0), 256, 2
    g = np.random.default_rng(11)
    start = g.uniform([0, 0], [SW, SH], size=(B, L, 1, 2))
    steps = g.normal(0, 1, size=(B, L, P, 2)) * [SW / P / 2, SH / P / 2] + [SW / P / 3, -SH / P / 4]
    pts = (start + np.cumsum(steps, axis=2)).astype(np.float32)
    npts = g.integers(0, P + 1, size=(B, L)).astype(np.int64)
    nlanes = g.integers(0, L + 1, size=(B,)).astype(np.int64)
    pts_d, npts_d, nlanes_d = torch.from_numpy(pts).to(DEV), torch.from_numpy(npts).to(DEV), torch.from_numpy(nlanes).to(DEV)
    shapes = [(B, int(SH / s), int(SW / s)) for s in strides]
    base = [torch.rand(s_, generator=torch.Generator().manual_seed(10 + i)).mul_(0.3) for i, s_ in enumerate(shapes)]
    maps = [t.to(DEV) for t in base]
    draw_polylines_multiscale(maps, pts_d, q_per_stride, radius, strides, 6.0, 0.9, num_points=npts_d,
                              num_lanes=nlanes_d, clear=clear)
    torch.cuda.synchronize()
    # oracle: arc-length-uniform samples (oracle/lane, pinned by tests/golden/lane_polyline.npz) -> int(p / stride)
    # -> oracle/h1.  A sample that the fp32 GPU sampler places within 1e-3 px of a pixel boundary of the stride grid
    # may legitimately round to the other side; such samples are counted and must be rare, and the map comparison
    # uses the GPU's own integer targets for them (stage-wise parity as in tests/test_lane_raster_gpu.py)
    from accvlab.draw_heatmap import sample_lane_targets

    frac = np.linspace(0.0, 1.0, q_per_stride).astype(np.float32)
    samples = np.full((B, L, q_per_stride, 2), np.nan, dtype=np.float64)
    for i in range(B):
        for j in range(L):
            n = int(npts[i, j])
            if n > 0:
                samples[i, j] = oracle_lane.sample(pts[i, j, :n], frac, relative=True)
    bad = np.isnan(samples).any(-1)
    for i, s in enumerate(strides):
        gc, gr = sample_lane_targets(pts_d, q_per_stride, radius, s, num_points=npts_d)
        gc = gc.cpu().numpy().reshape(B, L, q_per_stride, 2)
        gr = gr.cpu().numpy().reshape(B, L, q_per_stride)
        want_c = np.where(bad[..., None], 0, np.trunc(np.nan_to_num(samples) / s)).astype(np.int32)
        differs = (gc != want_c).any(-1) & ~bad
        near_edge = np.abs(np.nan_to_num(samples) / s - np.round(np.nan_to_num(samples) / s)).min(-1) < 1e-3 * max(SW, SH) / s / 100
        assert not (differs & ~near_edge).any(), "integer lane target differs away from a pixel boundary"
        assert differs.mean() < 1e-3
        assert np.array_equal(gr, np.where(bad, -1, radius))
        sizes = nlanes * q_per_stride
        want = base[i].numpy().copy()
        oracle_h1.draw_heatmap_batched(want, gc.reshape(B, -1, 2), gr.reshape(B, -1), sizes, k=0.9, clear=clear,
                                       threads=oracle_h1.max_threads())
        assert _max_err(maps[i], want) <= ATOL, f"stride {s}"


# ------------------------------------------------------------------------------------------------ > 2^31 elements
@pytest.mark.parametrize("api", ["batched", "flat"])
def test_classwise_map_above_2_31_elements(api):
    """INTEGRATION.md §3 claims 64-bit plane offsets where the reference's `int` arithmetic overflows
    (draw_heatmap_cuda_kernel.cuh:70,99,104).  64 x 20 x 1080 x 1920 = 2 654 208 000 elements (10.6 GB): objects in
    the first plane, a middle plane beyond the 2^31-element mark and the very last plane; every other plane must
    stay exactly zero."""
    from accvlab.draw_heatmap import draw_heatmap, draw_heatmap_batched

    B, C, H, W = 64, 20, 1080, 1920
    assert B * C * H * W > 2 ** 31
    free, _ = torch.cuda.mem_get_info(DEV)
    if free < 14 * 2 ** 30:
        pytest.skip("needs 11 GB of free device memory")
    nmax = 3
    centers = torch.zeros((B, nmax, 2), dtype=torch.int32)
    radii = torch.ones((B, nmax), dtype=torch.int32)
    labels = torch.zeros((B, nmax), dtype=torch.int32)
    counts = torch.zeros(B, dtype=torch.int64)
    placed = {(0, 0): [(100, 50, 40)], (51, 17): [(1900, 1070, 120), (7, 3, 9)], (63, 19): [(960, 540, 269), (1919, 1079, 33), (0, 1079, 5)]}
    assert 51 * C + 17 > 2 ** 31 // (H * W)                                     # beyond the int32 element offset
    for (s, cls), objs in placed.items():
        for j, (x, y, r) in enumerate(objs):
            centers[s, j, 0], centers[s, j, 1], radii[s, j], labels[s, j] = x, y, r, cls
        counts[s] = len(objs)
    hm = torch.empty((B, C, H, W), device=DEV)
    hm[0, 0].fill_(5.0)                                                          # junk in a touched and in an
    hm[63, 18].fill_(-2.0)                                                       # untouched plane: clear must remove it
    cd, rd, ld, nd = centers.to(DEV), radii.to(DEV), labels.to(DEV), counts.to(DEV)
    if api == "batched":
        rb = lambda t: SimpleNamespace(tensor=t, sample_sizes=nd)  # noqa: E731
        draw_heatmap_batched(hm, rb(cd), rb(rd), 6.0, 1.0, rb(ld), clear=True)
    else:
        valid = (torch.arange(nmax)[None, :] < counts[:, None])
        idx = (torch.arange(B, dtype=torch.int32)[:, None] * C + labels)[valid].contiguous()
        draw_heatmap(hm.view(B * C, H, W), centers[valid].contiguous().to(DEV), radii[valid].contiguous().to(DEV),
                     idx.to(DEV), 6.0, 1.0, clear=True)
    torch.cuda.synchronize()
    plane_max = hm.view(B * C, -1).amax(dim=1).cpu()
    plane_min = hm.view(B * C, -1).amin(dim=1).cpu()
    touched = sorted(s * C + cls for (s, cls) in placed)
    for p in range(B * C):
        if p in touched:
            assert float(plane_max[p]) == 1.0
        else:
            assert float(plane_max[p]) == 0.0 and float(plane_min[p]) == 0.0, f"plane {p} was written"
    for (s, cls), objs in placed.items():
        want = np.zeros((1, H, W), dtype=np.float32)
        c = np.array([[(x, y) for x, y, _ in objs]], dtype=np.int32)
        r = np.array([[r_ for _, _, r_ in objs]], dtype=np.int32)
        oracle_h1.draw_heatmap_batched(want, c, r, np.array([len(objs)]), clear=True)
        assert float(np.abs(hm[s, cls].cpu().numpy() - want[0]).max()) <= ATOL, (s, cls)
    # in-place semantics on the same huge map: max into the existing content, untouched planes keep theirs
    hm[63, 18].fill_(0.25)
    if api == "batched":
        draw_heatmap_batched(hm, rb(cd), rb(rd), 6.0, 0.5, rb(ld))
        torch.cuda.synchronize()
        assert float(hm[63, 18].min()) == 0.25 and float(hm[63, 18].max()) == 0.25
        assert float(hm[63, 19].max()) == 1.0                                    # k = 0.5 never exceeds the old peak
        assert float(hm[63, 19, 540, 960]) == 1.0
    del hm
    torch.cuda.empty_cache()
