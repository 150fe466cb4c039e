"""lane_helpers polyline ops: oracle pinned to golden vectors of the reference's own test oracle; CPU product path and
(marked gpu) the HIP kernel against the same vectors.  Tolerance 1e-5 abs as in the reference
(packages/lane_helpers/tests/polyline_test_utils.py:100,115); the golden values were computed in fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import lane as oracle

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lane_polyline.npz"), allow_pickle=False)
FIXED = ["rect", "batched", "deg", "one"] + [f"rand{k}" for k in range(12)]
RAGGED = [f"rag{k}" for k in range(6)]
ATOL = 1e-5


def _poly():
    from accvlab.lane_helpers import polyline
    return polyline


@pytest.mark.parametrize("name", FIXED)
def test_oracle_matches_reference_vectors_fixed(name):
    p, d, e = Z[f"{name}_points"], Z[f"{name}_dist"], Z[f"{name}_expected"]
    for b in range(p.shape[0]):
        got = oracle.sample(p[b], d[b])
        assert np.allclose(got, e[b], atol=ATOL, rtol=0)
        if f"{name}_length" in Z.files:
            assert abs(oracle.length(p[b]) - Z[f"{name}_length"][b]) <= 1e-4


@pytest.mark.parametrize("name", RAGGED)
def test_oracle_matches_reference_vectors_ragged(name):
    p, d, e = Z[f"{name}_points"], Z[f"{name}_dist"], Z[f"{name}_expected"]
    ps, qs, ln = Z[f"{name}_psizes"], Z[f"{name}_qsizes"], Z[f"{name}_length"]
    for b in range(p.shape[0]):
        got = oracle.sample(p[b, :ps[b]], d[b, :qs[b]])
        assert np.allclose(got, e[b, :qs[b]], atol=ATOL, rtol=0, equal_nan=True)
        assert np.isnan(ln[b]) == (ps[b] == 0)
        if ps[b] > 0:
            assert abs(oracle.length(p[b, :ps[b]]) - ln[b]) <= 1e-4


def _run_fixed(dev, name, relative):
    poly = _poly()
    p, d, e = torch.from_numpy(Z[f"{name}_points"]), torch.from_numpy(Z[f"{name}_dist"]), Z[f"{name}_expected"]
    din = d
    if relative:
        tot = torch.linalg.vector_norm(p[:, 1:] - p[:, :-1], dim=2).sum(1) if p.shape[1] > 1 else torch.zeros(p.shape[0])
        if float(tot.min()) <= 0:
            return
        din = d / tot[:, None]
    got = poly.interpolate(p.to(dev), din.to(dev), relative=relative)
    assert got.shape == e.shape and got.dtype == p.dtype and got.device.type == torch.device(dev).type
    assert np.allclose(got.cpu().numpy(), e, atol=2e-5 if relative else ATOL, rtol=0)
    if f"{name}_length" in Z.files:
        ln = poly.lengths(p.to(dev))
        assert np.allclose(ln.cpu().numpy(), Z[f"{name}_length"], atol=1e-4, rtol=0)


def _run_ragged(dev, name, idt):
    from accvlab.batching_helpers import RaggedBatch

    poly = _poly()
    p, d = torch.from_numpy(Z[f"{name}_points"]), torch.from_numpy(Z[f"{name}_dist"])
    ps, qs = torch.from_numpy(Z[f"{name}_psizes"]).to(idt), torch.from_numpy(Z[f"{name}_qsizes"]).to(idt)
    prb = RaggedBatch(p.to(dev), sample_sizes=ps.to(dev))
    drb = RaggedBatch(d.to(dev), sample_sizes=qs.to(dev))
    res = poly.interpolate_var_size_batch(prb, drb)
    assert isinstance(res, RaggedBatch) and torch.equal(res.sample_sizes.cpu(), qs)
    e = Z[f"{name}_expected"]
    for b in range(p.shape[0]):
        n = int(qs[b])
        assert np.allclose(res.tensor[b, :n].cpu().numpy(), e[b, :n], atol=ATOL, rtol=0, equal_nan=True), (name, b)
    ln = poly.lengths_var_size_batch(prb).cpu().numpy()
    assert np.allclose(ln, Z[f"{name}_length"], atol=1e-4, rtol=0, equal_nan=True)


@pytest.mark.parametrize("relative", [False, True])
@pytest.mark.parametrize("name", FIXED)
def test_cpu_fixed(name, relative):
    _run_fixed("cpu", name, relative)


@pytest.mark.parametrize("name", RAGGED)
def test_cpu_ragged(name):
    _run_ragged("cpu", name, torch.int64)


def test_cpu_edge_cases_and_validation():
    poly = _poly()
    empty = poly.interpolate(torch.empty((2, 0, 3)), torch.tensor([[0.0, 1.0], [-1.0, 2.0]]))
    assert empty.shape == (2, 2, 3) and bool(torch.isnan(empty).all())
    assert poly.interpolate(torch.empty((2, 0, 3)), torch.empty((2, 0))).shape == (2, 0, 3)
    assert bool(torch.isnan(poly.lengths(torch.empty((2, 0, 3)))).all())
    assert poly.lengths(torch.ones((2, 1, 3))).tolist() == [0.0, 0.0]
    nc = torch.tensor([[[0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 2.0, 2.0]]]).transpose(1, 2)   # non-contiguous input
    out = poly.interpolate(nc, torch.tensor([[0.5, 2.0]]))
    assert np.allclose(out.numpy(), [[[0.5, 0.0], [1.0, 1.0]]], atol=1e-6)
    with pytest.raises(RuntimeError):
        poly.interpolate(torch.zeros(2, 3), torch.zeros(2, 1))
    with pytest.raises(RuntimeError):
        poly.interpolate(torch.zeros(2, 3, 2), torch.zeros(3, 1))
    with pytest.raises(RuntimeError):
        poly.interpolate(torch.zeros(2, 3, 2), torch.zeros(2, 1, dtype=torch.float64))


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("relative", [False, True])
@pytest.mark.parametrize("name", FIXED)
def test_gpu_fixed(name, relative):
    _run_fixed("cuda:0", name, relative)


@pytest.mark.gpu
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
@pytest.mark.parametrize("name", RAGGED)
def test_gpu_ragged(name, idt):
    _run_ragged("cuda:0", name, idt)


@pytest.mark.gpu
def test_gpu_dtypes_long_polylines_and_edge_cases():
    poly = _poly()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    # long polylines: LDS path (8k points) and the global-scratch path (20k points > 48 KB of fp32)
    for npnt in (8192, 20000):
        p = torch.rand((3, npnt, 2), generator=g, dtype=torch.float64)
        tot = torch.linalg.vector_norm(p[:, 1:] - p[:, :-1], dim=2).sum(1)
        d = torch.rand((3, 300), generator=g, dtype=torch.float64) * tot[:, None]
        ref = np.stack([oracle.sample(p[b].numpy(), d[b].numpy()) for b in range(3)])
        got64 = poly.interpolate(p.to(dev), d.to(dev))
        assert np.allclose(got64.cpu().numpy(), ref, atol=1e-9, rtol=0)
        got32 = poly.interpolate(p.float().to(dev), d.float().to(dev))
        assert np.abs(got32.cpu().numpy() - ref).max() < 5e-3      # fp32 accumulation over ~1e4 segments
        assert np.allclose(poly.lengths(p.to(dev)).cpu().numpy(), tot.numpy(), rtol=1e-12)
    p = torch.rand((4, 40, 3), generator=g)
    tot = torch.linalg.vector_norm(p[:, 1:] - p[:, :-1], dim=2).sum(1)
    d = torch.rand((4, 25), generator=g) * tot[:, None]
    ref = np.stack([oracle.sample(p[b].numpy(), d[b].numpy()) for b in range(4)])
    for dt, tol in ((torch.float16, 2e-2), (torch.bfloat16, 1e-1)):
        got = poly.interpolate(p.to(dev, dt), d.to(dev, dt))
        assert got.dtype == dt and np.abs(got.float().cpu().numpy() - ref).max() < tol
    empty = poly.interpolate(torch.empty((2, 0, 3), device=dev), torch.tensor([[0.0, 1.0], [-1.0, 2.0]], device=dev))
    assert empty.shape == (2, 2, 3) and bool(torch.isnan(empty).all())
    assert poly.interpolate(torch.empty((2, 0, 3), device=dev), torch.empty((2, 0), device=dev)).shape == (2, 0, 3)
    assert bool(torch.isnan(poly.lengths(torch.empty((2, 0, 3), device=dev))).all())
    assert poly.lengths(torch.ones((2, 1, 3), device=dev)).tolist() == [0.0, 0.0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("relative", [False, True])
def test_native_host_path_matches_the_torch_formulation_and_the_oracle(dtype, relative):
    """CPU tensors run accv_polyline_sample_host (C++, double accumulation — the reference's polyline_cpu.cpp:28-132);
    cross-checked against the float64 torch formulation kept in ops.py and, per polyline, against oracle/lane.py, on
    ragged batches with empty polylines, repeated points (zero-length segments) and queries outside [0, length]."""
    from accvlab.lane_helpers.polyline import ops
    from oracle import lane as oracle_lane

    g = torch.Generator().manual_seed(5)
    b, p, q, d = 300, 17, 23, 3
    pts = torch.cumsum(torch.rand(b, p, d, generator=g, dtype=torch.float64), 1).to(dtype)
    pts[:, 5] = pts[:, 4]                                          # a zero-length segment in every polyline
    n = torch.randint(0, p + 1, (b,), generator=g)
    n[:3] = torch.tensor([0, 1, p])
    m = torch.randint(0, q + 1, (b,), generator=g)
    dist = (torch.rand(b, q, generator=g, dtype=torch.float64) * (1.4 if relative else 12.0) - (0.2 if relative else 1.0)).to(dtype)
    got, _ = ops._host(pts, dist, n, m, relative, True, False)
    ref = ops._cpu_interpolate(pts, dist, n, m, relative)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for i in range(b):
        k = int(m[i])
        a, r = got[i, :k].double(), ref[i, :k].double()
        assert torch.equal(torch.isnan(a), torch.isnan(r))
        assert float((torch.nan_to_num(a) - torch.nan_to_num(r)).abs().max() if k else 0.0) <= tol * 20
        if i < 40:
            want = oracle_lane.sample(pts[i, :int(n[i])].numpy(), dist[i, :k].numpy(), relative=relative)
            assert np.allclose(np.nan_to_num(a.numpy()), np.nan_to_num(want), atol=tol * 20, rtol=0)
    _, lens = ops._host(pts, None, n, None, False, False, True)
    ref_l = ops._cpu_lengths(pts, n)
    assert torch.equal(torch.isnan(lens), torch.isnan(ref_l)) and float(torch.nan_to_num(lens - ref_l).abs().max()) <= tol * 20
    assert bool(torch.isnan(lens[0])) and float(lens[1]) == 0.0     # empty -> NaN, single point -> 0


@pytest.mark.gpu
def test_cpp_host_path_and_python_path_agree_and_unusual_inputs_fall_back(monkeypatch):
    """interpolate / lengths on CUDA tensors go through csrc_host/lane_host.cpp; it must give what the python formulation of the
    same call gives (bit for bit: same kernel), and decline — not mis-handle — what only the python path covers: non-contiguous
    tensors, mismatching dtypes (RuntimeError with the reference's message)"""
    import torch

    from accvlab import _amd_native as nat
    from accvlab.lane_helpers.polyline import interpolate, lengths, ops

    if nat.NO_HOST_FASTPATH:
        pytest.skip("ACCV_NO_HOST_FASTPATH=1: the suite is running over the python formulations only")
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    for dt in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        pts = (torch.rand(5, 9, 2, generator=g) * 10).to(dt).to(dev)
        dist = (torch.rand(5, 7, generator=g) * 12).to(dt).to(dev)
        assert ops._native() is not None, "build the host extensions (make -C accv-lab_amd/csrc_host)"
        fast_i, fast_l = interpolate(pts, dist), lengths(pts)
        fast_r = interpolate(pts, dist, relative=True)
        with monkeypatch.context() as m:
            m.setattr(ops, "_lh", None)
            assert torch.equal(interpolate(pts, dist), fast_i) and torch.equal(lengths(pts), fast_l)
            assert torch.equal(interpolate(pts, dist, relative=True), fast_r)
        # non-contiguous points: declined by the C++ path, handled by the python one
        wide = (torch.rand(5, 9, 4, generator=g) * 10).to(dt).to(dev)
        nc = wide[:, :, :2]
        assert not nc.is_contiguous() and torch.equal(interpolate(nc, dist), interpolate(nc.contiguous(), dist))
    with pytest.raises(RuntimeError, match="same dtype"):
        interpolate(torch.rand(2, 3, 2, device=dev), torch.rand(2, 4, device=dev, dtype=torch.float64))
    with pytest.raises(RuntimeError):
        interpolate(torch.rand(2, 3, 2, device=dev), torch.rand(3, 4, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("npnt,nq", [(10, 3000), (600, 5000), (3000, 4100), (2048, 9000)])
def test_gpu_many_queries_are_chunked_over_workgroups(npnt, nq):
    """few polylines x thousands of queries: the queries of a polyline are spread over several workgroups (each repeats the
    scan; 1024-thread workgroups from 2048 points on) — fixed and ragged, relative, and the group boxes written alongside"""
    from accvlab.batching_helpers import RaggedBatch
    from accvlab.draw_heatmap import sample_lanes

    poly = _poly()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(npnt + nq)
    b = 3
    p = torch.rand((b, npnt, 2), generator=g, dtype=torch.float64) * 3
    tot = torch.linalg.vector_norm(p[:, 1:] - p[:, :-1], dim=2).sum(1)
    d = (torch.rand((b, nq), generator=g, dtype=torch.float64) * 1.2 - 0.1) * tot[:, None]
    ref = np.stack([oracle.sample(p[i].numpy(), d[i].numpy()) for i in range(b)])
    got = poly.interpolate(p.to(dev), d.to(dev))
    assert np.allclose(got.cpu().numpy(), ref, atol=1e-9, rtol=0)
    rel = poly.interpolate(p.to(dev), (d / tot[:, None]).to(dev), relative=True)
    assert np.allclose(rel.cpu().numpy(), ref, atol=1e-8, rtol=0)
    n_p = torch.tensor([npnt, max(1, npnt // 3), 0])
    n_q = torch.tensor([nq, nq // 2 + 7, nq - 1])
    rg = poly.interpolate_var_size_batch(RaggedBatch(p.to(dev), sample_sizes=n_p.to(dev)), RaggedBatch(d.to(dev), sample_sizes=n_q.to(dev)))
    for i in range(b):
        want = oracle.sample(p[i, : n_p[i]].numpy(), d[i, : n_q[i]].numpy())
        assert np.allclose(rg.tensor[i, : n_q[i]].cpu().numpy(), want, atol=1e-9, rtol=0, equal_nan=True)
    # the sampler of the lane raster with its group boxes (float32, 2-D): one box per 64 consecutive samples
    q = (nq // 64) * 64
    lanes = p.float().view(1, b, npnt, 2).to(dev)
    boxes = torch.empty(1 * ((b * q + 63) // 64) * 4 + 4, device=dev)
    samples = sample_lanes(lanes, q, group_boxes_ptr=boxes.data_ptr())
    torch.cuda.synchronize()
    grp = samples.view(b * q // 64, 64, 2)
    want_boxes = torch.cat([grp.min(1).values, grp.max(1).values], 1)
    assert torch.equal(boxes[: want_boxes.numel()].view(-1, 4), want_boxes)
