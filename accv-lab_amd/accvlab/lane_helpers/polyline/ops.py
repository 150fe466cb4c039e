"""Batched polyline interpolation / lengths — same four functions, argument meaning and validation as the reference
(packages/lane_helpers/accvlab/lane_helpers/polyline/functions.py:27-111 over ext_impl/polyline/src/polyline.cpp:101-398).

CUDA tensors run ONE kernel of libaccv_hip.so per call (accv_polyline_sample); CPU tensors use a vectorised torch
implementation that accumulates in float64 (the reference's CPU path accumulates in at::acc_type, polyline_cpu.cpp).
"""
from __future__ import annotations

import functools

import torch

from ... import _amd_native as _nat

try:  # C++ fast path of the fixed-size CUDA calls (built by `make -C accv-lab_amd/csrc_host`); the code below is complete without it
    from . import _lane_host as _lh
except ImportError:  # pragma: no cover
    _lh = None
_native_bound = False


def _native():
    """_lane_host with the C-ABI addresses bound (same library handle as the ctypes binding), or None."""
    global _native_bound
    if _lh is None or _nat.NO_HOST_FASTPATH:
        return None
    if not _native_bound:
        import ctypes

        h = _nat.ctypes_lib()
        addr = lambda name: ctypes.cast(getattr(h, name), ctypes.c_void_p).value  # noqa: E731
        _lh.bind_native(addr("accv_polyline_sample"), addr("accv_polyline_scratch_bytes"), addr("accv_last_error"))
        _native_bound = True
    return _lh

_DTYPE_CODE = {torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3}


def _req(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_points(points, name="points"):
    # messages as check_device / check_type of the reference (ext_impl/polyline/src/polyline.cpp:40-63)
    if not (isinstance(points, torch.Tensor)):
        raise RuntimeError(f"{name} must be a tensor")
    if not (points.device.type in ("cpu", "cuda")):
        raise RuntimeError(f"{name} must be a CPU or CUDA tensor")
    if points.is_cuda:
        if points.dtype not in _DTYPE_CODE:
            raise RuntimeError(f"{name} must have dtype float16, float32, float64, or bfloat16 on CUDA")
    elif points.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"{name} must have dtype float32 or float64 on CPU")


def _check_sizes(sizes, limit, name):
    if not (sizes.dtype in (torch.int32, torch.int64)):
        raise RuntimeError(f"{name} must have dtype int32 or int64")
    if not (sizes.dim() == 1):
        raise RuntimeError(f"{name} must be a 1D tensor")


def _check_size_values(checks):
    """Every count must lie in [0, padded extent] (check_sample_sizes, polyline.cpp:74-81).  The reference reads two
    flags back per size tensor; here all of a call's size tensors are checked with ONE read-back (none for CPU tensors
    beyond the comparison itself).  The kernels clamp anyway, so internal callers that build the counts themselves
    (accvlab.draw_heatmap.lanes) skip this."""
    bad = None
    for sizes, limit, _ in checks:
        if sizes.numel():
            b = ((sizes < 0) | (sizes > limit)).any()
            bad = b if bad is None else (bad | b.to(bad.device))
    if bad is None or not bool(bad):
        return
    for sizes, limit, name in checks:
        if sizes.numel() and bool(((sizes < 0) | (sizes > limit)).any()):
            raise RuntimeError(f"{name} values must be in [0, {limit}]")


@functools.lru_cache(maxsize=256)
def _scratch_bytes(batch: int, max_points: int, code: int) -> int:
    """accv_polyline_scratch_bytes, remembered per shape (a size_t result goes through ctypes: ~3 us per call)"""
    return int(_nat.lib().accv_polyline_scratch_bytes(batch, max_points, code))


def _gpu(points, distances, p_sizes, d_sizes, relative, want_points, want_lengths, group_boxes_ptr=0):
    """``group_boxes_ptr``: device pointer for the bounding boxes of every 64 consecutive samples (f32, 2-D points only;
    accv_polyline_sample_boxes) or 0."""
    lib = _nat.lib()
    b, pmax, dims = points.shape
    qmax = distances.shape[1] if distances is not None else 0
    code = _DTYPE_CODE[points.dtype]
    out_p = torch.empty((b, qmax, dims), dtype=points.dtype, device=points.device) if want_points else None
    out_l = torch.empty((b,), dtype=points.dtype, device=points.device) if want_lengths else None
    if b == 0 or (want_points and not want_lengths and (qmax == 0 or dims == 0)):
        return out_p, out_l
    points = points.contiguous()
    distances = distances.contiguous() if distances is not None else None
    # ONE width flag covers both count arrays in the C-ABI: take it from whichever is present and convert the other if
    # it differs (an int64 array read as int32 would silently truncate the number of samples)
    c64 = 0
    present = [t for t in (p_sizes, d_sizes) if t is not None]
    if present:
        wide = any(t.dtype == torch.int64 for t in present)
        want = torch.int64 if wide else torch.int32
        c64 = 1 if wide else 0
        if p_sizes is not None:
            p_sizes = p_sizes.to(want).contiguous()
        if d_sizes is not None:
            d_sizes = d_sizes.to(want).contiguous()
    with _nat.device_guard(points.device):
        sb = _scratch_bytes(b, pmax, code)
        scratch = torch.empty(sb, dtype=torch.uint8, device=points.device) if sb else None
        _nat.check(lib.accv_polyline_sample_boxes(
            points.data_ptr(), distances.data_ptr() if distances is not None else None,
            p_sizes.data_ptr() if p_sizes is not None else None, d_sizes.data_ptr() if d_sizes is not None else None,
            out_p.data_ptr() if out_p is not None else None, out_l.data_ptr() if out_l is not None else None,
            group_boxes_ptr or None, b, pmax, qmax, dims, code, c64, int(bool(relative)),
            scratch.data_ptr() if scratch is not None else None, sb, _nat.stream_ptr(points.device)), "polyline")
    return out_p, out_l


def _host(points, distances, p_sizes, d_sizes, relative, want_points, want_lengths):
    """CPU tensors: the native host implementation behind the C-ABI (accv_polyline_sample_host — double accumulation and
    threading as the reference's polyline_cpu.cpp:28-132).  float32 / float64; the torch formulations below
    (_cpu_interpolate / _cpu_lengths) remain as its cross-check in the tests."""
    lib = _nat.lib()
    b, pmax, dims = points.shape
    qmax = distances.shape[1] if distances is not None else 0
    out_p = torch.empty((b, qmax, dims), dtype=points.dtype) if want_points else None
    out_l = torch.empty((b,), dtype=points.dtype) if want_lengths else None
    if b == 0:
        return out_p, out_l
    points = points.contiguous()
    distances = distances.contiguous() if distances is not None else None
    c64 = 0
    present = [t for t in (p_sizes, d_sizes) if t is not None]
    if present:
        wide = any(t.dtype == torch.int64 for t in present)
        want = torch.int64 if wide else torch.int32
        c64 = 1 if wide else 0
        p_sizes = p_sizes.to(want).contiguous() if p_sizes is not None else None
        d_sizes = d_sizes.to(want).contiguous() if d_sizes is not None else None
    _nat.check(lib.accv_polyline_sample_host(
        points.data_ptr(), distances.data_ptr() if distances is not None else None,
        p_sizes.data_ptr() if p_sizes is not None else None, d_sizes.data_ptr() if d_sizes is not None else None,
        out_p.data_ptr() if out_p is not None and out_p.numel() else None, out_l.data_ptr() if out_l is not None else None,
        b, pmax, qmax, dims, _DTYPE_CODE[points.dtype], c64, int(bool(relative)), 0), "polyline (host)")
    return out_p, out_l


def _cpu_accum(points, p_sizes):
    """float64 accumulated distances [B, P] with +inf behind the valid points, and total lengths."""
    b, pmax, _ = points.shape
    acc = torch.zeros((b, max(pmax, 1)), dtype=torch.float64)
    n = torch.full((b,), pmax, dtype=torch.int64) if p_sizes is None else p_sizes.to(torch.int64).clamp(0, pmax)
    if pmax > 1:
        seg = torch.linalg.vector_norm(points[:, 1:].double() - points[:, :-1].double(), dim=2)
        valid = torch.arange(pmax - 1).unsqueeze(0) < (n - 1).unsqueeze(1)
        acc[:, 1:pmax] = torch.cumsum(seg * valid, dim=1)
    total = acc.gather(1, (n - 1).clamp(min=0).unsqueeze(1)).squeeze(1)
    return acc[:, :pmax] if pmax > 0 else acc[:, :0], n, total


def _cpu_interpolate(points, distances, p_sizes, d_sizes, relative):
    b, pmax, dims = points.shape
    q = distances.shape[1]
    out = torch.empty((b, q, dims), dtype=points.dtype)
    if b == 0 or q == 0 or dims == 0:
        return out
    acc, n, total = _cpu_accum(points, p_sizes)
    d = distances.double()
    if relative:
        d = d * total.unsqueeze(1)
    if pmax == 0:
        return out.fill_(float("nan"))
    inf_tail = torch.arange(pmax).unsqueeze(0) >= n.unsqueeze(1)
    keys = acc.masked_fill(inf_tail, float("inf"))
    idx = torch.searchsorted(keys, d.contiguous(), right=True) - 1            # last index with accum <= d
    last = (n - 1).clamp(min=0).unsqueeze(1)
    lo = idx.clamp(min=0).minimum(last)
    hi = (lo + 1).minimum(last)
    d0, d1 = acc.gather(1, lo), acc.gather(1, hi)
    seg = d1 - d0
    eps = torch.finfo(torch.float64).eps
    w1 = torch.where(seg >= eps, (d - d0) / seg.clamp(min=eps), torch.zeros_like(d))
    w1 = torch.where((idx < 0) | (idx >= last), torch.zeros_like(w1), w1)     # clamped to an end point
    p0 = points.double().gather(1, lo.unsqueeze(-1).expand(b, q, dims))
    p1 = points.double().gather(1, hi.unsqueeze(-1).expand(b, q, dims))
    res = p0 * (1.0 - w1).unsqueeze(-1) + p1 * w1.unsqueeze(-1)
    res = torch.where((w1 == 0).unsqueeze(-1), p0, res)                       # exact end / lower points
    res = torch.where((n == 0).view(b, 1, 1), torch.full_like(res, float("nan")), res)
    out.copy_(res)
    if d_sizes is not None:
        pass  # entries behind distances.sample_sizes are unspecified padding
    return out


def _cpu_lengths(points, p_sizes):
    b = points.shape[0]
    if b == 0:
        return torch.empty((0,), dtype=points.dtype)
    _, n, total = _cpu_accum(points, p_sizes)
    total = torch.where(n == 0, torch.full_like(total, float("nan")), total)
    return total.to(points.dtype)


def interpolate(points: torch.Tensor, distances: torch.Tensor, *, relative: bool = False) -> torch.Tensor:
    """Sample every polyline of ``points (batch, num_points, num_dims)`` at ``distances (batch, num_distances)`` measured
    along the polyline from its first point (``relative=True``: as fractions of its total length).  Queries before the
    start / beyond the end clamp to the first / last point.  Returns ``(batch, num_distances, num_dims)``."""
    if isinstance(points, torch.Tensor) and points.is_cuda and isinstance(distances, torch.Tensor):
        native = _native()      # plain CUDA call: checks, allocation and the launch in C++ (declines anything unusual)
        if native is not None:
            res = native.interpolate(points, distances, bool(relative))
            if res is not None:
                return res
    _check_points(points)
    _check_points(distances, "distances")
    if not (points.dim() == 3):
        raise RuntimeError("points must have shape (batch, num_points, num_dims)")
    if not (distances.dim() == 2):
        raise RuntimeError("distances must have shape (batch, num_distances)")
    if not (points.size(0) == distances.size(0)):
        raise RuntimeError("points and distances must contain the same number of polylines")
    if not (points.dtype == distances.dtype):
        raise RuntimeError("points and distances must have the same dtype")
    if not (points.device == distances.device):
        raise RuntimeError("points and distances must be on the same device")
    if points.is_cuda:
        return _gpu(points, distances, None, None, relative, True, False)[0]
    return _host(points, distances, None, None, relative, True, False)[0]


def lengths(points: torch.Tensor) -> torch.Tensor:
    """Total length of every polyline of ``points (batch, num_points, num_dims)`` -> ``(batch,)``."""
    if isinstance(points, torch.Tensor) and points.is_cuda:
        native = _native()
        if native is not None:
            res = native.lengths(points)
            if res is not None:
                return res
    _check_points(points)
    if not (points.dim() == 3):
        raise RuntimeError("points must have shape (batch, num_points, num_dims)")
    if points.is_cuda:
        return _gpu(points, None, None, None, False, False, True)[1]
    return _host(points, None, None, None, False, False, True)[1]


def _check_var(points, sizes, name):
    _check_sizes(sizes, points.size(1), name)
    if not (sizes.device == points.device):
        raise RuntimeError(f"{name} must be on the same device as its tensor")
    if not (sizes.size(0) == points.size(0)):
        raise RuntimeError(f"{name} must contain one count per polyline")


def interpolate_var_size_batch(points, distances, *, relative: bool = False):
    """Ragged variant: ``points`` / ``distances`` are RaggedBatch-like (``(batch, max_num_points, num_dims)`` and
    ``(batch, max_num_distances)``, one batch dimension, non-uniform dimension 1).  Returns a RaggedBatch with the
    distances' sample sizes."""
    assert points.num_batch_dims == 1, "points must have exactly one batch dimension"
    assert distances.num_batch_dims == 1, "distances must have exactly one batch dimension"
    assert points.non_uniform_dim == 1, "points.non_uniform_dim must be 1 for shape (batch, max_num_points, num_dims)"
    assert distances.non_uniform_dim == 1, "distances.non_uniform_dim must be 1 for shape (batch, max_num_distances)"
    pt, dt, ps, ds = points.tensor, distances.tensor, points.sample_sizes, distances.sample_sizes
    _check_points(pt)
    _check_points(dt, "distances")
    if not (pt.dim() == 3):
        raise RuntimeError("points must have shape (batch, max_num_points, num_dims)")
    if not (dt.dim() == 2):
        raise RuntimeError("distances must have shape (batch, max_num_distances)")
    if not (pt.size(0) == dt.size(0)):
        raise RuntimeError("points and distances must contain the same number of polylines")
    if not (pt.dtype == dt.dtype):
        raise RuntimeError("points and distances must have the same dtype")
    if not (pt.device == dt.device):
        raise RuntimeError("points and distances must be on the same device")
    _req(ps.dtype == ds.dtype, "points.sample_sizes and distances.sample_sizes must have the same dtype "
                               "(both int32 or both int64)")
    _check_var(pt, ps, "points.sample_sizes")
    _check_var(dt, ds, "distances.sample_sizes")
    _check_size_values([(ps, pt.size(1), "points.sample_sizes"), (ds, dt.size(1), "distances.sample_sizes")])
    if pt.is_cuda:
        res = _gpu(pt, dt, ps, ds, relative, True, False)[0]
    else:
        res = _host(pt, dt, ps, ds, relative, True, False)[0]
    return distances.create_with_sample_sizes_like_self(res)


def lengths_var_size_batch(points) -> torch.Tensor:
    """Total length of every polyline of a ragged batch -> ``(batch,)``."""
    assert points.num_batch_dims == 1, "points must have exactly one batch dimension"
    assert points.non_uniform_dim == 1, "points.non_uniform_dim must be 1 for shape (batch, max_num_points, num_dims)"
    pt, ps = points.tensor, points.sample_sizes
    _check_points(pt)
    if not (pt.dim() == 3):
        raise RuntimeError("points must have shape (batch, max_num_points, num_dims)")
    _check_var(pt, ps, "points.sample_sizes")
    _check_size_values([(ps, pt.size(1), "points.sample_sizes")])
    if pt.is_cuda:
        return _gpu(pt, None, ps, None, False, False, True)[1]
    return _host(pt, None, ps, None, False, False, True)[1]
