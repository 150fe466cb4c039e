"""GPU parity tests of the HIP rasteriser (through the python operator API -> C-ABI) against
(a) the committed golden vectors of the reference's python oracle and (b) the CPU oracle on seeded inputs.

Tolerance: 1e-5 absolute on fp32 heat-map values (BASELINE.json north_star); the reference's own bar is
MSE < 1e-3 (packages/draw_heatmap/tests/test_draw_heatmap.py:85)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import bench_workloads as wl
import h1_cases
from oracle import h1 as oracle

pytestmark = pytest.mark.gpu
ATOL = 1e-5
DEV = "cuda:0"


def rb(t, sizes):
    return SimpleNamespace(tensor=t, sample_sizes=sizes)


_VARIANTS = {"shipped": 0, "rows16": "HM_TILE_ROWS_16", "rows16-wt-nt": ("HM_TILE_ROWS_16", "HM_WRITE_THROUGH"),
             "rows8-wt-nt": ("HM_TILE_ROWS_8", "HM_WRITE_THROUGH"), "small-splat": "HM_SMALL_RADII",
             "plain-stores": "HM_PLAIN_STORES"}


@pytest.fixture(autouse=True, params=list(_VARIANTS), ids=list(_VARIANTS))
def kernel_variant(request):
    """every test of this module runs against the default dispatch and against every other splat-kernel instantiation
    the shipped library holds, selected through the PUBLIC hint flags (tile rows, write-through stores, the box-walking
    small-splat kernel); the dispatch actually taken is checked"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import ops

    names = _VARIANTS[request.param]
    names = () if names == 0 else ((names,) if isinstance(names, str) else names)
    flags = 0
    for n in names:
        flags |= getattr(nat, n)
    ops._FORCED_FLAGS = flags
    yield request.param
    ops._FORCED_FLAGS = 0


def _dh():
    from accvlab.draw_heatmap import draw_heatmap, draw_heatmap_batched
    return draw_heatmap, draw_heatmap_batched


def _close(got: torch.Tensor, exp: np.ndarray, what=""):
    g = got.detach().cpu().numpy()
    assert g.shape == exp.shape
    err = float(np.nanmax(np.abs(g.astype(np.float64) - exp.astype(np.float64)))) if g.size else 0.0
    assert err <= ATOL, f"{what}: max abs err {err}"
    return err


def _flat_from_padded(centers, radii, sizes, labels=None, C=0):
    cs, rs, idx = [], [], []
    for s, n in enumerate(sizes.tolist()):
        cs.append(centers[s, :n])
        rs.append(radii[s, :n])
        idx.append(np.full(n, s, dtype=np.int32) if labels is None else (s * C + labels[s, :n]).astype(np.int32))
    return np.concatenate(cs), np.concatenate(rs), np.concatenate(idx)


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


@pytest.mark.parametrize("gname", ["h1_g1.npz", "h1_g2.npz"])
@pytest.mark.parametrize("clear", [False, True])
def test_golden_batched_flat_classwise(gname, clear):
    draw_heatmap, draw_heatmap_batched = _dh()
    z = h1_cases.load(gname)
    k, factor, C = float(z["k"]), float(z["factor"]), int(z["C"])
    exp = z["expected"]
    B, H, W = exp.shape
    centers, radii, sizes, labels = _t(z["centers"]), _t(z["radii"]), _t(z["sizes"]), _t(z["labels"])
    fill = 123.0 if clear else 0.0  # clear=True must ignore previous content
    # batched, int64 sample sizes
    hm = torch.full((B, H, W), fill, device=DEV)
    draw_heatmap_batched(hm, rb(centers, sizes), rb(radii, sizes), factor, k, clear=clear)
    _close(hm, exp, "batched")
    # batched, int32 sample sizes
    hm = torch.full((B, H, W), fill, device=DEV)
    draw_heatmap_batched(hm, rb(centers, sizes.to(torch.int32)), rb(radii, sizes), factor, k, clear=clear)
    _close(hm, exp, "batched i32 sizes")
    # flat
    c, r, idx = _flat_from_padded(z["centers"], z["radii"], z["sizes"])
    hm = torch.full((B, H, W), fill, device=DEV)
    draw_heatmap(hm, _t(c), _t(r), _t(idx), factor, k, clear=clear)
    _close(hm, exp, "flat")
    # class-wise
    cw = torch.full((B, C, H, W), fill, device=DEV)
    draw_heatmap_batched(cw, rb(centers, sizes), rb(radii, sizes), factor, k, labels=rb(labels, sizes), clear=clear)
    if "cw_planes" in z.files:
        full = np.zeros((B, C, H, W), dtype=np.float32)
        for (s, cl), e in zip(z["cw_planes"].tolist(), z["cw_expected"]):
            full[s, cl] = e
    else:
        full = z["cw_expected"]
    _close(cw, full, "class-wise")
    # flat drawing into B*C planes == class-wise
    c, r, idx = _flat_from_padded(z["centers"], z["radii"], z["sizes"], z["labels"], C)
    fl = torch.full((B * C, H, W), fill, device=DEV)
    draw_heatmap(fl, _t(c), _t(r), _t(idx), factor, k, clear=clear)
    _close(fl.view(B, C, H, W), full, "flat class planes")


@pytest.mark.parametrize("case", list(h1_cases.g3_cases()), ids=lambda c: c[0])
def test_golden_edge_cases(case):
    draw_heatmap, draw_heatmap_batched = _dh()
    name, H, W, c, r, k, factor, base, exp = case
    n = len(r)
    hm = torch.full((1, H, W), base, device=DEV)
    draw_heatmap(hm, _t(c.reshape(-1, 2)), _t(r), torch.zeros(n, dtype=torch.int32, device=DEV), factor, k)
    _close(hm[0], exp, f"flat {name}")
    hm = torch.full((1, H, W), base, device=DEV)
    sizes = torch.tensor([n], device=DEV)
    draw_heatmap_batched(hm, rb(_t(c.reshape(1, -1, 2)), sizes), rb(_t(r.reshape(1, -1)), sizes), factor, k)
    _close(hm[0], exp, f"batched {name}")
    if base == 0.0 and k > 0:
        hm = torch.full((1, H, W), 9.0, device=DEV)
        draw_heatmap_batched(hm, rb(_t(c.reshape(1, -1, 2)), sizes), rb(_t(r.reshape(1, -1)), sizes), factor, k,
                             clear=True)
        _close(hm[0], exp, f"batched clear {name}")


@pytest.mark.parametrize("frame", list(h1_cases.g4_frames()), ids=lambda f: f[0])
def test_golden_full_hd(frame):
    _, draw_heatmap_batched = _dh()
    name, H, W, T, c, r, crc, sums, picks, tiles = frame
    sizes = torch.tensor([len(r)], device=DEV)
    hm = torch.empty((1, H, W), device=DEV)
    draw_heatmap_batched(hm, rb(_t(c[None]), sizes), rb(_t(r[None]), sizes), clear=True)
    got = hm[0].cpu().numpy()
    tx = W // T
    for p, t in zip(picks.tolist(), tiles):
        i, j = p // tx, p % tx
        assert np.abs(got[i * T:(i + 1) * T, j * T:(j + 1) * T] - t).max() <= ATOL
    _, got_sums = h1_cases.tile_crc_and_sums(got, T)
    # every tile: |sum diff| <= T*T*ATOL, in practice ~1e-4
    assert np.abs(got_sums - sums).max() <= T * T * ATOL
    # and the whole frame against the CPU oracle
    ref = np.zeros((1, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(ref, c[None], r[None], np.array([len(r)]), clear=True, threads=8)
    err = _close(hm, ref, f"full frame {name}")
    print(f"{name}: max abs err vs oracle {err:.3e}")


@pytest.mark.parametrize("H,W,rule", [(1080, 1920, "A"), (270, 480, "A"), (135, 240, "B"), (67, 129, "A"),
                                      (540, 960, "B")])
@pytest.mark.parametrize("clear", [False, True])
def test_random_batches_vs_oracle(H, W, rule, clear):
    _, draw_heatmap_batched = _dh()
    B = 6
    cl, rl = wl.heatmap_objects(B, H, W, 0, 128, rule, seed=H + W)
    cpad, sizes = wl.pad_ragged(cl, 2)  # padding holds a drawable object at (2,2)
    rpad, _ = wl.pad_ragged(rl, 3)
    base = np.random.RandomState(1).rand(B, H, W).astype(np.float32) * 0.3 - 0.1
    ref = base.copy()
    oracle.draw_heatmap_batched(ref, cpad.numpy(), rpad.numpy(), sizes.numpy(), k=0.9, clear=clear, threads=8)
    hm = _t(base)
    draw_heatmap_batched(hm, rb(cpad.to(DEV), sizes.to(DEV)), rb(rpad.to(DEV), sizes), 6.0, 0.9, clear=clear)
    _close(hm, ref, f"{H}x{W} rule {rule} clear={clear}")


def test_classwise_random_and_bad_labels():
    _, draw_heatmap_batched = _dh()
    B, C, H, W = 5, 7, 96, 160
    cl, rl, ll = wl.heatmap_objects(B, H, W, 0, 40, "A", seed=5, n_classes=C)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl, 1)
    lpad, _ = wl.pad_ragged(ll)
    if sizes[0] > 1:
        lpad[0, 0] = -1      # out-of-range labels are ignored (the reference device-asserts)
        lpad[0, 1] = C + 3
    ref = np.zeros((B, C, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(ref, cpad.numpy(), rpad.numpy(), sizes.numpy(), labels=lpad.numpy())
    hm = torch.zeros((B, C, H, W), device=DEV)
    draw_heatmap_batched(hm, rb(cpad.to(DEV), sizes.to(DEV)), rb(rpad.to(DEV), sizes), labels=rb(lpad.to(DEV), sizes))
    _close(hm, ref, "class-wise random")


def test_flat_many_objects_unsorted_and_out_of_range_planes():
    draw_heatmap, _ = _dh()
    P, H, W, N = 37, 64, 128, 5000
    g = np.random.RandomState(3)
    c = np.stack([g.randint(-5, W + 5, N), g.randint(-5, H + 5, N)], 1).astype(np.int32)
    r = g.randint(0, 12, N).astype(np.int32)
    idx = g.randint(-2, P + 2, N).astype(np.int32)
    ref = np.zeros((P, H, W), dtype=np.float32)
    oracle.draw_heatmap_flat(ref, c, r, idx, 6.0, 1.0)
    hm = torch.zeros((P, H, W), device=DEV)
    draw_heatmap(hm, _t(c), _t(r), _t(idx))
    _close(hm, ref, "flat many")


@pytest.mark.parametrize("P,H,W,N", [(9000, 8, 16, 20000),     # more planes than the single-launch binning handles
                                     (5, 32, 64, 70000)])      # more objects than it handles
def test_flat_large_counts_take_the_multi_kernel_binning(P, H, W, N):
    draw_heatmap, _ = _dh()
    g = np.random.RandomState(4)
    c = np.stack([g.randint(-3, W + 3, N), g.randint(-3, H + 3, N)], 1).astype(np.int32)
    r = g.randint(0, 4, N).astype(np.int32)
    idx = g.randint(-1, P + 1, N).astype(np.int32)
    ref = np.zeros((P, H, W), dtype=np.float32)
    oracle.draw_heatmap_flat(ref, c, r, idx, 6.0, 1.0)
    hm = torch.zeros((P, H, W), device=DEV)
    draw_heatmap(hm, _t(c), _t(r), _t(idx))
    _close(hm, ref, "flat large counts")


def test_idempotent_and_monotone():
    """size-independent properties at BASELINE's full size: drawing twice changes nothing; in-place draw on
    a zero map equals the fused clear; result >= base everywhere."""
    _, draw_heatmap_batched = _dh()
    B, H, W = 8, 1080, 1920
    cl, rl = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=42)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl)
    c, r = rb(cpad.to(DEV), sizes.to(DEV)), rb(rpad.to(DEV), sizes.to(DEV))
    a = torch.zeros((B, H, W), device=DEV)
    draw_heatmap_batched(a, c, r)
    b = torch.full((B, H, W), -7.0, device=DEV)
    draw_heatmap_batched(b, c, r, clear=True)
    assert torch.equal(a, b)
    a2 = a.clone()
    draw_heatmap_batched(a2, c, r)
    assert torch.equal(a, a2)
    assert float(a.max()) <= 1.0 and float(a.min()) >= 0.0
    # every object's centre pixel holds exactly k (exp(0) == 1)
    for s in range(B):
        n = int(sizes[s])
        xy = cpad[s, :n].long()
        ok = (xy[:, 0] >= 0) & (xy[:, 0] < W) & (xy[:, 1] >= 0) & (xy[:, 1] < H)
        assert torch.all(a[s][xy[ok, 1].to(DEV), xy[ok, 0].to(DEV)] == 1.0)


def test_empty_and_degenerate_inputs():
    draw_heatmap, draw_heatmap_batched = _dh()
    hm = torch.ones((2, 8, 12), device=DEV)
    e2 = torch.zeros((0, 2), dtype=torch.int32, device=DEV)
    e1 = torch.zeros((0,), dtype=torch.int32, device=DEV)
    draw_heatmap(hm, e2, e1, e1)
    assert torch.all(hm == 1)
    draw_heatmap(hm, e2, e1, e1, clear=True)
    assert torch.all(hm == 0)
    sizes = torch.zeros(2, dtype=torch.int64, device=DEV)
    hm = torch.ones((2, 8, 12), device=DEV)
    draw_heatmap_batched(hm, rb(torch.zeros((2, 0, 2), dtype=torch.int32, device=DEV), sizes),
                         rb(torch.zeros((2, 0), dtype=torch.int32, device=DEV), sizes))
    assert torch.all(hm == 1)
    # zero-sized map
    draw_heatmap_batched(torch.zeros((2, 0, 12), device=DEV),
                         rb(torch.zeros((2, 3, 2), dtype=torch.int32, device=DEV), sizes),
                         rb(torch.zeros((2, 3), dtype=torch.int32, device=DEV), sizes))
    # unaligned base pointer (odd element offset) falls back to the scalar-store variant
    big = torch.zeros(1 + 2 * 16 * 64, device=DEV)
    view = big[1:].view(2, 16, 64)
    sizes1 = torch.tensor([1, 1], device=DEV)
    cen = torch.tensor([[[5, 5]], [[60, 10]]], dtype=torch.int32, device=DEV)
    rad = torch.tensor([[3], [6]], dtype=torch.int32, device=DEV)
    draw_heatmap_batched(view, rb(cen, sizes1), rb(rad, sizes1))
    ref = np.zeros((2, 16, 64), dtype=np.float32)
    oracle.draw_heatmap_batched(ref, cen.cpu().numpy(), rad.cpu().numpy(), np.array([1, 1]))
    _close(view, ref, "unaligned view")
    assert float(big[0]) == 0.0


def test_error_behaviour_matches_reference():
    draw_heatmap, draw_heatmap_batched = _dh()
    hm = torch.zeros((2, 8, 8), device=DEV)
    c = torch.zeros((3, 2), dtype=torch.int32, device=DEV)
    r = torch.ones((3,), dtype=torch.int32, device=DEV)
    i = torch.zeros((3,), dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        draw_heatmap(hm.cpu(), c, r, i)                     # not a CUDA tensor
    with pytest.raises(RuntimeError):
        draw_heatmap(hm.transpose(1, 2), c, r, i)           # not contiguous
    with pytest.raises(RuntimeError):
        draw_heatmap(hm, c, r[:2], i)                       # dim0 mismatch
    with pytest.raises(RuntimeError):
        draw_heatmap(hm, c.to(torch.int64), r, i)           # wrong dtype
    with pytest.raises(RuntimeError):
        draw_heatmap(hm.double(), c, r, i)                  # fp32 only
    with pytest.raises(RuntimeError):
        draw_heatmap(hm[0], c, r, i)                        # rank
    sizes = torch.tensor([1, 1], device=DEV)
    cb = torch.zeros((2, 3, 2), dtype=torch.int32, device=DEV)
    rbt = torch.ones((2, 3), dtype=torch.int32, device=DEV)
    with pytest.raises(AssertionError):
        draw_heatmap_batched(hm, rb(cb, sizes), rb(rbt[:, :2], sizes))
    with pytest.raises(AssertionError):
        draw_heatmap_batched(hm, rb(cb[:1], sizes), rb(rbt, sizes))
    with pytest.raises(RuntimeError):
        draw_heatmap_batched(hm[:1], rb(cb, sizes), rb(rbt, sizes))      # batch mismatch
    with pytest.raises(RuntimeError):
        draw_heatmap_batched(hm, rb(cb, sizes), rb(rbt, sizes), labels=rb(rbt, sizes))  # needs rank-4 map


def test_runs_on_current_stream_without_sync():
    _, draw_heatmap_batched = _dh()
    B, H, W = 4, 256, 512
    cl, rl = wl.heatmap_objects(B, H, W, 1, 64, "A", seed=9)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl)
    ref = np.zeros((B, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(ref, cpad.numpy(), rpad.numpy(), sizes.numpy())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        hm = torch.zeros((B, H, W), device=DEV)
        c, r = rb(cpad.to(DEV, non_blocking=True), sizes.to(DEV)), rb(rpad.to(DEV), sizes.to(DEV))
        draw_heatmap_batched(hm, c, r)
    side.synchronize()
    _close(hm, ref, "side stream")


def test_target_prep_front_end_gpu_bit_exact_and_multi_scale():
    """Fused bbox -> (centre, radius) kernel: bit-exact against the reference helper's outputs stored in G2, and a
    C3-style multi-scale run (strides 4/8/16) against the CPU path + oracle."""
    from accvlab.batching_helpers import RaggedBatch
    from accvlab.draw_heatmap import draw_heatmap_batched, get_centers_and_radii

    z = h1_cases.load("h1_g2.npz")
    c, r = get_centers_and_radii(_t(z["centers_f"]), _t(z["boxes_f"]), int(z["stride"]))
    assert np.array_equal(c.cpu().numpy(), z["centers"]) and np.array_equal(r.cpu().numpy(), z["radii"])
    g = torch.Generator().manual_seed(5)
    B, SH, SW = 4, 432, 768
    cf, bf = [], []
    for _ in range(B):
        n = int(torch.randint(0, 40, (1,), generator=g))
        cc = torch.rand(n, 2, generator=g) * torch.tensor([SW, SH])
        half = torch.rand(n, 4, generator=g) * 90
        cf.append(cc)
        bf.append(torch.cat([cc - half[:, :2], cc + half[:, 2:]], 1))
    cpad, sizes = wl.pad_ragged(cf)
    bpad, _ = wl.pad_ragged(bf)
    crb = RaggedBatch(cpad.to(DEV), sample_sizes=sizes.to(DEV))
    brb = RaggedBatch(bpad.to(DEV), sample_sizes=sizes.to(DEV))
    for s in (4, 8, 16):
        ci, ri = get_centers_and_radii(crb, brb, s)
        ci_cpu, ri_cpu = get_centers_and_radii(cpad, bpad, s)
        assert isinstance(ci, RaggedBatch) and torch.equal(ci.tensor.cpu(), ci_cpu) and torch.equal(ri.tensor.cpu(), ri_cpu)
        hm = torch.empty((B, SH // s, SW // s), device=DEV)
        draw_heatmap_batched(hm, ci, ri, clear=True)
        ref = np.zeros((B, SH // s, SW // s), dtype=np.float32)
        oracle.draw_heatmap_batched(ref, ci_cpu.numpy(), ri_cpu.numpy(), sizes.numpy(), clear=True)
        _close(hm, ref, f"multi-scale stride {s}")


def test_full_size_equivalences_and_scaling():
    """BASELINE-sized (1080x1920) properties that need no oracle: the three entry points are bitwise identical on the
    same objects (max is order independent and they share the per-object arithmetic); the map scales with k; adding
    objects never lowers a pixel."""
    draw_heatmap, draw_heatmap_batched = _dh()
    B, H, W = 6, 1080, 1920
    cl, rl = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=77)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl)
    c, r = rb(cpad.to(DEV), sizes.to(DEV)), rb(rpad.to(DEV), sizes.to(DEV))
    a = torch.empty((B, H, W), device=DEV)
    draw_heatmap_batched(a, c, r, clear=True)
    # flat with shuffled object order
    cf, rf, idx = _flat_from_padded(cpad.numpy(), rpad.numpy(), sizes.numpy())
    perm = np.random.RandomState(0).permutation(len(rf))
    f = torch.zeros((B, H, W), device=DEV)
    draw_heatmap(f, _t(cf[perm]), _t(rf[perm]), _t(idx[perm]))
    assert torch.equal(a, f)
    # class-wise with every label == 2 lands in plane 2 only
    labels = rb(torch.full_like(rpad, 2).to(DEV), sizes.to(DEV))
    cw = torch.empty((B, 3, H, W), device=DEV)
    draw_heatmap_batched(cw, c, r, labels=labels, clear=True)
    assert torch.equal(cw[:, 2], a) and float(cw[:, :2].abs().max()) == 0.0
    # scaling in k
    half = torch.empty_like(a)
    draw_heatmap_batched(half, c, r, 6.0, 0.5, clear=True)
    assert float((half - 0.5 * a).abs().max()) <= 1e-6
    # monotone in the object set: drawing only the first half of every sample is <= the full map
    part = torch.empty_like(a)
    draw_heatmap_batched(part, rb(c.tensor, (sizes // 2).to(DEV)), rb(r.tensor, (sizes // 2).to(DEV)), clear=True)
    assert bool((part <= a).all())
    # and drawing the rest on top of it in place reproduces the full map
    rest_c = [t[n // 2:] for t, n in zip(cl, sizes.tolist())]
    rest_r = [t[n // 2:] for t, n in zip(rl, sizes.tolist())]
    rc, rs = wl.pad_ragged(rest_c)
    rr, _ = wl.pad_ragged(rest_r)
    draw_heatmap_batched(part, rb(rc.to(DEV), rs.to(DEV)), rb(rr.to(DEV), rs.to(DEV)))
    assert torch.equal(part, a)


def test_batched_op_is_graph_capturable():
    """The batched entry point enqueues exactly one kernel, allocates nothing and never synchronises: it can be
    captured into a hipGraph (torch.cuda.CUDAGraph) and replayed with updated inputs."""
    _, draw_heatmap_batched = _dh()
    B, H, W = 3, 128, 256
    cl, rl = wl.heatmap_objects(B, H, W, 1, 20, "A", seed=11)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl)
    c, r = rb(cpad.to(DEV), sizes.to(DEV)), rb(rpad.to(DEV), sizes.to(DEV))
    hm = torch.zeros((B, H, W), device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        draw_heatmap_batched(hm, c, r, clear=True)       # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        draw_heatmap_batched(hm, c, r, clear=True)
    ref = np.zeros((B, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(ref, cpad.numpy(), rpad.numpy(), sizes.numpy(), clear=True)
    hm.fill_(5.0)
    graph.replay()
    _close(hm, ref, "graph replay")
    # new object data in the SAME buffers, replay again
    cl2, rl2 = wl.heatmap_objects(B, H, W, 1, 20, "A", seed=12)
    c2, s2 = wl.pad_ragged(cl2)
    r2, _ = wl.pad_ragged(rl2)
    n = min(c2.shape[1], cpad.shape[1])
    s2 = s2.clamp(max=n)
    c.tensor[:, :n].copy_(c2[:, :n].to(DEV))
    r.tensor[:, :n].copy_(r2[:, :n].to(DEV))
    c.sample_sizes.copy_(s2.to(DEV))
    graph.replay()
    ref2 = np.zeros((B, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(ref2, c.tensor.cpu().numpy(), r.tensor.cpu().numpy(), s2.numpy(), clear=True)
    _close(hm, ref2, "graph replay with new inputs")


def test_one_round_launches_take_the_taller_tiles():
    """a fused-clear launch whose 128 x 16 tiles are more than the chip holds at once while its 128 x 32 tiles all fit (the
    8-frame shards of the strong-scaling split) runs with the taller tiles — one round of resident tiles; same map either way"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import ops

    _, draw_heatmap_batched = _dh()
    pinned, ops._FORCED_FLAGS = ops._FORCED_FLAGS, 0      # (the kernel-variant fixture pins the tile height: lifted for this test)
    try:
        _one_round_rule_cases(nat, draw_heatmap_batched)
    finally:
        ops._FORCED_FLAGS = pinned


def _one_round_rule_cases(nat, draw_heatmap_batched):
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    g = torch.Generator().manual_seed(11)
    for frames, clear, want16 in ((8, True, True), (16, True, False), (8, False, False), (4, True, False)):
        h, w = 1080, 1920
        tiles8, tiles16 = frames * 15 * 68, frames * 15 * 34
        assert want16 == (clear and tiles8 > 24 * cus and tiles16 <= 16 * cus), "the test's table assumes 256 compute units"
        c = torch.stack([torch.randint(0, w, (frames, 40), generator=g), torch.randint(0, h, (frames, 40), generator=g)], -1).int().to(DEV)
        r = torch.randint(1, 40, (frames, 40), generator=g).int().to(DEV)
        n = torch.randint(1, 41, (frames,), generator=g).to(DEV)
        base = torch.rand(frames, h, w, generator=g).mul_(0.2).to(DEV)
        got = base.clone()
        draw_heatmap_batched(got, rb(c, n), rb(r, n), clear=clear)
        assert ("R=16" in nat.last_dispatch()) == want16, (frames, clear, nat.last_dispatch())
        ref = base.clone()
        draw_heatmap_batched(ref, rb(c, n), rb(r, n), clear=clear, tile_rows=8)
        assert "R=8" in nat.last_dispatch()
        assert torch.equal(got, ref)


def test_dispatch_string_follows_the_public_hints(kernel_variant):
    """accv_draw_heatmap_last_dispatch reports the instantiation that ran (bench.py derives roofline.kernel from it)"""
    from accvlab import _amd_native as nat

    _, draw_heatmap_batched = _dh()
    hm = torch.zeros(2, 64, 256, device=DEV)
    c = torch.tensor([[[10, 10]], [[100, 30]]], dtype=torch.int32, device=DEV)
    r = torch.tensor([[3], [5]], dtype=torch.int32, device=DEV)
    n = torch.tensor([1, 1], device=DEV)
    draw_heatmap_batched(hm, rb(c, n), rb(r, n), clear=True)
    s = nat.last_dispatch()
    if kernel_variant == "small-splat":
        assert s.startswith("splat_small_kernel<")
    else:
        assert s.startswith("splat_kernel<PX=4,")
        assert ("R=16" in s) == kernel_variant.startswith("rows16")
    assert ("SM=4" in s) == kernel_variant.endswith("wt-nt")
    assert "CLEAR=1" in s and "block(64)" in s
    # in-place launches choose their stores per plane unless a hint fixes the policy
    draw_heatmap_batched(hm, rb(c, n), rb(r, n))
    s = nat.last_dispatch()
    if kernel_variant != "small-splat":
        want = "SM=4" if kernel_variant.endswith("wt-nt") else "SM=0" if kernel_variant == "plain-stores" else "SM=5"
        assert want in s and "CLEAR=0" in s
    else:
        # the small-splat kernel has no per-plane choice: point-like objects are the sparse case, i.e. plain stores
        # unless write-through is asked for explicitly
        assert s.startswith("splat_small_kernel<") and "SM=0" in s and "CLEAR=0" in s
    # a map whose width is not a multiple of 4 takes the scalar instantiation whatever the hints say
    odd = torch.zeros(1, 8, 30, device=DEV)
    draw_heatmap_batched(odd, rb(c[:1], n[:1]), rb(r[:1], n[:1]))
    assert nat.last_dispatch().startswith("splat_kernel<PX=1,R=8,CLEAR=0,SM=0>")
    with pytest.raises(RuntimeError):
        draw_heatmap_batched(hm, rb(c, n), rb(r, n), tile_rows=12)


def test_time_next_launch_brackets_one_kernel_and_is_one_shot():
    """accv_draw_heatmap_time_next_launch: the next splat launch of this thread carries start/stop events (same kernel, same
    result), the one after it does not"""
    import ctypes

    from accvlab import _amd_native as nat

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    hip.hipEventQuery.argtypes = [ctypes.c_void_p]
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    ev = []
    for _ in range(2):
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        ev.append(e.value)
    _, draw_heatmap_batched = _dh()
    c = torch.tensor([[[10, 10], [200, 50]], [[100, 30], [5, 90]]], dtype=torch.int32)
    r = torch.tensor([[3, 20], [5, 7]], dtype=torch.int32)
    sizes = torch.tensor([2, 2])
    cd, rd, sd = c.to(DEV), r.to(DEV), sizes.to(DEV)
    want = torch.empty(2, 96, 256, device=DEV)
    draw_heatmap_batched(want, rb(cd, sd), rb(rd, sd), clear=True)
    got = torch.empty_like(want)
    nat.check(nat.lib().accv_draw_heatmap_time_next_launch(ev[0], ev[1]), "time_next_launch")
    draw_heatmap_batched(got, rb(cd, sd), rb(rd, sd), clear=True)
    torch.cuda.synchronize()
    ms = ctypes.c_float(-1.0)
    assert hip.hipEventElapsedTime(ctypes.byref(ms), ev[0], ev[1]) == 0 and 0.0 < ms.value < 50.0
    assert torch.equal(got, want)
    # one-shot: a further launch leaves the recorded interval untouched
    draw_heatmap_batched(got, rb(cd, sd), rb(rd, sd), clear=True)
    torch.cuda.synchronize()
    ms2 = ctypes.c_float(-1.0)
    assert hip.hipEventElapsedTime(ctypes.byref(ms2), ev[0], ev[1]) == 0 and ms2.value == ms.value
    for e in ev:
        hip.hipEventDestroy(e)


@pytest.mark.parametrize("clear", [True, False])
def test_classwise_call_without_object_slots(clear):
    """labels of shape [B, 0] have no storage (null data pointer): a class-wise call with Nmax == 0 must not be rejected —
    it clears (clear=True) or leaves (in-place) the planes, as the reference's launch over zero targets does"""
    _, draw_heatmap_batched = _dh()
    hm = torch.full((2, 3, 16, 64), 0.25, device=DEV)
    c = torch.zeros((2, 0, 2), dtype=torch.int32, device=DEV)
    r = torch.zeros((2, 0), dtype=torch.int32, device=DEV)
    n = torch.zeros(2, dtype=torch.int64, device=DEV)
    draw_heatmap_batched(hm, rb(c, n), rb(r, n), labels=rb(torch.zeros((2, 0), dtype=torch.int32, device=DEV), n), clear=clear)
    assert float(hm.min()) == float(hm.max()) == (0.0 if clear else 0.25)


def test_cpp_host_path_of_draw_heatmap_batched_agrees_with_the_python_path(monkeypatch):
    """plain draw_heatmap_batched calls go through csrc_host/dh_host.cpp: same map as the python formulation of the call (batched
    and class-wise, clear and in-place, int32 and int64 counts), and inputs it does not take (non-contiguous objects, a counts
    dtype that needs a cast) reach the python path instead of being mis-handled"""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import ops

    if nat.NO_HOST_FASTPATH:
        pytest.skip("ACCV_NO_HOST_FASTPATH=1: the suite is running over the python formulations only")
    assert ops._native() is not None, "build the host extensions (make -C accv-lab_amd/csrc_host)"
    _, draw_heatmap_batched = _dh()
    g = torch.Generator().manual_seed(9)
    b, n, h, w, ncls = 3, 7, 40, 72, 4
    c = torch.stack([torch.randint(0, w, (b, n), generator=g), torch.randint(0, h, (b, n), generator=g)], -1).to(torch.int32).to(DEV)
    r = torch.randint(0, 9, (b, n), generator=g).to(torch.int32).to(DEV)
    lab = torch.randint(0, ncls, (b, n), generator=g).to(torch.int32).to(DEV)
    for cnt in (torch.tensor([7, 0, 3]), torch.tensor([7, 0, 3], dtype=torch.int32), torch.tensor([7, 0, 3], dtype=torch.int16)):
        cnt = cnt.to(DEV)
        for clear in (True, False):
            base, base_cw = torch.rand(b, h, w, generator=g).to(DEV) * 0.3, torch.rand(b, ncls, h, w, generator=g).to(DEV) * 0.3
            fast, fast_cw = base.clone(), base_cw.clone()
            draw_heatmap_batched(fast, rb(c, cnt), rb(r, cnt), 6.0, 0.9, clear=clear)
            draw_heatmap_batched(fast_cw, rb(c, cnt), rb(r, cnt), 6.0, 0.9, labels=rb(lab, cnt), clear=clear)
            with monkeypatch.context() as m:
                m.setattr(ops, "_dh", None)
                slow, slow_cw = base.clone(), base_cw.clone()
                draw_heatmap_batched(slow, rb(c, cnt), rb(r, cnt), 6.0, 0.9, clear=clear)
                draw_heatmap_batched(slow_cw, rb(c, cnt), rb(r, cnt), 6.0, 0.9, labels=rb(lab, cnt), clear=clear)
            assert torch.equal(fast, slow) and torch.equal(fast_cw, slow_cw)
    with pytest.raises(RuntimeError):        # non-contiguous centres: declined in C++, diagnosed by the python checks
        wide = torch.zeros(b, n, 4, dtype=torch.int32, device=DEV)
        draw_heatmap_batched(torch.zeros(b, h, w, device=DEV), rb(wide[:, :, :2], cnt.to(torch.int64)), rb(r, cnt.to(torch.int64)))


@pytest.mark.parametrize("clear", [True, False])
def test_flat_call_single_launch_and_binned_paths_agree(clear, monkeypatch):
    """draw_heatmap with few objects runs as ONE launch (the flat input as a one-sample class-wise call); with many it bins the
    objects per plane first.  Same maps, incl. objects whose plane index is out of range (ignored by both)."""
    from accvlab import _amd_native as nat
    from accvlab.draw_heatmap import ops

    draw_heatmap, _ = _dh()
    g = torch.Generator().manual_seed(31)
    p, h, w, n = 7, 48, 200, 300
    c = torch.stack([torch.randint(-5, w + 5, (n,), generator=g), torch.randint(-5, h + 5, (n,), generator=g)], -1).to(torch.int32).to(DEV)
    r = torch.randint(-1, 12, (n,), generator=g).to(torch.int32).to(DEV)
    idx = torch.randint(-2, p + 2, (n,), generator=g).to(torch.int32).to(DEV)
    base = (torch.rand(p, h, w, generator=g) * 0.3).to(DEV)
    direct = base.clone()
    draw_heatmap(direct, c, r, idx, 6.0, 0.8, clear=clear)
    kernel_direct = nat.last_dispatch()
    monkeypatch.setattr(ops, "_FLAT_DIRECT_MAX_OBJECTS", 0)
    binned = base.clone()
    draw_heatmap(binned, c, r, idx, 6.0, 0.8, clear=clear)
    assert torch.equal(direct, binned)
    assert f",{p}) block(64)" in kernel_direct                      # one plane per "class" of the single sample
    want = base.cpu().numpy().copy()
    if clear:
        want[:] = 0
    oracle.draw_heatmap_flat(want, c.cpu().numpy(), r.cpu().numpy(), idx.cpu().numpy(), 6.0, 0.8)
    _close(direct, want, "flat, single launch")
