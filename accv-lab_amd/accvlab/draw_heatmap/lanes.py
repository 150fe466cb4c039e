"""Lane (polyline) rasteriser — the "lane_helpers polyline raster" of BASELINE.json config 3.

EXTENSION: the reference ships no polyline rasteriser; its lane package only samples polylines
(packages/lane_helpers/accvlab/lane_helpers/polyline/functions.py:27-111).  A lane is drawn here by composing the two
reference operators that exist: sample every lane at ``num_samples`` arc-length-uniform positions (``interpolate`` with
``relative=True``) and splat every sample as a Gaussian of a fixed radius (``draw_heatmap_batched`` with the
``small_radii`` hint: the splat kernel that walks each sample's few-pixel box).  Three launches, no host synchronisation:

    accv_polyline_sample  ->  accv_heatmap_targets_from_points_f32  ->  accv_draw_heatmap_batched_f32
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from .. import _amd_native as _nat
from ..lane_helpers.polyline import ops as _poly
from . import ops as _ops
from .ops import draw_heatmap_batched

_cache: dict = {}


def _cached(key, make):
    """Device constants (sample fractions, full-size counts) shared by every later call — on ANY stream.  The tensor is
    created on the stream that is current at the first call, so that stream is drained once before the tensor is
    published; while a stream is being captured into a graph nothing is cached (no synchronisation allowed there)."""
    t = _cache.get(key)
    if t is None:
        t = make()
        if torch.cuda.is_current_stream_capturing():
            return t
        torch.cuda.current_stream(t.device).synchronize()
        if len(_cache) > 64:
            _cache.clear()
        _cache[key] = t
    return t


def _fraction_rows(rows: int, num_samples: int, dev) -> torch.Tensor:
    """Arc-length fractions of the samples, one row per polyline: k / (num_samples - 1) in IEEE float32 (computed on the host so
    that the fused lane raster, which forms the same quotient inside its kernel, lands on the same samples bit for bit)."""
    if num_samples > 1:
        row = np.arange(num_samples, dtype=np.float32) / np.float32(num_samples - 1)
    else:
        row = np.zeros(1, np.float32)
    return torch.from_numpy(row).to(dev).unsqueeze(0).expand(rows, num_samples).contiguous()


def sample_lane_targets(polylines: torch.Tensor, num_samples: int, radius: int, out_size_factor: float = 1.0, *,
                        num_points: Optional[torch.Tensor] = None):
    """``polylines`` f32 ``[B, L, P, 2]`` (x, y in source pixels; ``num_points`` int ``[B, L]`` = valid points per
    lane, default all) -> ``(centers i32 [B, L*num_samples, 2], radii i32 [B, L*num_samples])`` at the heat-map stride
    ``out_size_factor``: ``c = int(sample / stride)``.  Samples of empty lanes get radius -1 (never drawn)."""
    if not (isinstance(polylines, torch.Tensor) and polylines.is_cuda):
        raise RuntimeError("polylines must be a CUDA tensor")
    if not (polylines.dim() == 4 and polylines.size(3) == 2):
        raise RuntimeError("polylines must be of shape [batch, lanes, points, 2]")
    if not (polylines.dtype == torch.float32):
        raise RuntimeError(f"polylines: expected float32 but found {polylines.dtype}")
    if not (num_samples >= 1):
        raise RuntimeError("num_samples must be >= 1")
    b, l, p, _ = polylines.shape
    dev = polylines.device
    pts = polylines.contiguous().view(b * l, p, 2)
    counts = None
    if num_points is not None:
        if not (num_points.shape == (b, l)):
            raise RuntimeError("num_points must be of shape [batch, lanes]")
        if not (num_points.device == dev):
            raise RuntimeError("num_points must be on the same device as polylines")
        _poly._check_sizes(num_points.reshape(-1), p, "num_points")
        counts = num_points.contiguous().view(b * l)
    centers = torch.empty((b, l * num_samples, 2), dtype=torch.int32, device=dev)
    radii = torch.empty((b, l * num_samples), dtype=torch.int32, device=dev)
    if b * l == 0:
        return centers, radii
    # arc-length fractions 0..1, one row per lane (a cached constant: no per-call kernel)
    frac = _cached(("frac", b * l, num_samples, dev), lambda: _fraction_rows(b * l, num_samples, dev))
    samples = _poly._gpu(pts, frac, counts, None, True, True, False)[0]
    with _nat.device_guard(dev):
        _nat.check(_nat.lib().accv_heatmap_targets_from_points_f32(
            samples.data_ptr(), b * l * num_samples, float(out_size_factor), int(radius), centers.data_ptr(),
            radii.data_ptr(), _nat.stream_ptr(dev)), "sample_lane_targets")
    return centers, radii


def draw_polylines_batched(heatmap: torch.Tensor, polylines: torch.Tensor, num_samples: int, radius: int,
                           out_size_factor: float = 1.0, diameter_to_sigma_factor: float = 6.0, k_scale: float = 1.0,
                           *, num_points: Optional[torch.Tensor] = None, num_lanes: Optional[torch.Tensor] = None,
                           clear: bool = False) -> None:
    """Draw ``polylines`` f32 ``[B, L, P, 2]`` into ``heatmap`` f32 ``[B, H, W]`` (in place, element-wise max; with
    ``clear=True`` fused zero-fill + draw) as chains of Gaussians of ``radius``.

    ``num_points`` int ``[B, L]``: valid points per lane; ``num_lanes`` int ``[B]``: only the first ``num_lanes[b]``
    lanes of a frame are drawn (both default to "all").  Choose ``num_samples`` so that the sample spacing
    (lane length / stride / (num_samples-1)) stays below ``radius`` for a gap-free line.

    Radii of a few pixels on an aligned map take the two-launch path of :func:`draw_polylines_multiscale` with this one
    scale (sampler + group boxes, then the point splat with its two-level cull: stride 4 of config 3 ≈ 19 + 6 µs against
    30 + 6 + 4 µs for sampler -> integer targets -> ``draw_heatmap_batched``); the results are bit-identical
    (``tests/test_lane_raster_gpu.py`` compares the two formulations)."""
    if (0 <= radius <= 7 and isinstance(heatmap, torch.Tensor) and heatmap.is_cuda and heatmap.dim() == 3
            and heatmap.is_contiguous() and heatmap.dtype == torch.float32 and heatmap.size(2) % 4 == 0
            and heatmap.data_ptr() % 16 == 0 and heatmap.size(1) * heatmap.size(2) * 4 < (1 << 31)
            and isinstance(polylines, torch.Tensor) and polylines.dim() == 4 and heatmap.size(0) == polylines.size(0)
            and heatmap.device == polylines.device):
        draw_polylines_multiscale([heatmap], polylines, num_samples, radius, [out_size_factor], diameter_to_sigma_factor,
                                  k_scale, num_points=num_points, num_lanes=num_lanes, clear=clear)
        return
    _draw_polylines_via_targets(heatmap, polylines, num_samples, radius, out_size_factor, diameter_to_sigma_factor, k_scale,
                                num_points=num_points, num_lanes=num_lanes, clear=clear)


def _draw_polylines_via_targets(heatmap: torch.Tensor, polylines: torch.Tensor, num_samples: int, radius: int,
                                out_size_factor: float = 1.0, diameter_to_sigma_factor: float = 6.0, k_scale: float = 1.0,
                                *, num_points: Optional[torch.Tensor] = None, num_lanes: Optional[torch.Tensor] = None,
                                clear: bool = False) -> None:
    """The composition of the reference operators spelled out: sampler -> integer targets -> ``draw_heatmap_batched`` (three
    launches; any radius, any map alignment).  The parity tests pin the lane raster stage by stage on this formulation."""
    centers, radii = sample_lane_targets(polylines, num_samples, radius, out_size_factor, num_points=num_points)
    b, l = polylines.shape[:2]
    if num_lanes is None:
        sizes = _cached(("full", b, l * num_samples, heatmap.device),
                        lambda: torch.full((b,), l * num_samples, dtype=torch.int32, device=heatmap.device))
    else:
        if not (num_lanes.shape == (b,)):
            raise RuntimeError("num_lanes must be of shape [batch]")
        sizes = num_lanes.clamp(0, l) * num_samples
    draw_heatmap_batched(heatmap, SimpleNamespace(tensor=centers, sample_sizes=sizes),
                         SimpleNamespace(tensor=radii, sample_sizes=sizes), diameter_to_sigma_factor, k_scale,
                         clear=clear, small_radii=radius <= 7)


def sample_lanes(polylines: torch.Tensor, num_samples: int, *, num_points: Optional[torch.Tensor] = None,
                 group_boxes_ptr: int = 0) -> torch.Tensor:
    """Arc-length-uniform samples of ``polylines`` f32 ``[B, L, P, 2]`` -> f32 ``[B, L * num_samples, 2]`` (source
    pixels; samples of empty lanes are NaN).  One launch of the polyline sampler; with ``group_boxes_ptr`` (and
    ``num_samples % 64 == 0``) the same launch also writes the bounding box of every 64 consecutive samples there."""
    if not (isinstance(polylines, torch.Tensor) and polylines.is_cuda):
        raise RuntimeError("polylines must be a CUDA tensor")
    if not (polylines.dim() == 4 and polylines.size(3) == 2):
        raise RuntimeError("polylines must be of shape [batch, lanes, points, 2]")
    if not (polylines.dtype == torch.float32):
        raise RuntimeError(f"polylines: expected float32 but found {polylines.dtype}")
    if not (num_samples >= 1):
        raise RuntimeError("num_samples must be >= 1")
    b, l, p, _ = polylines.shape
    dev = polylines.device
    if b * l == 0:
        return torch.empty((b, l * num_samples, 2), dtype=torch.float32, device=dev)
    counts = None
    if num_points is not None:
        if not (num_points.shape == (b, l) and num_points.device == dev):
            raise RuntimeError("num_points must be of shape [batch, lanes] on the polylines' device")
        _poly._check_sizes(num_points.reshape(-1), p, "num_points")
        counts = num_points.contiguous().view(b * l)
    frac = _cached(("frac", b * l, num_samples, dev), lambda: _fraction_rows(b * l, num_samples, dev))
    samples = _poly._gpu(polylines.contiguous().view(b * l, p, 2), frac, counts, None, True, True, False,
                         group_boxes_ptr)[0]
    return samples.view(b, l * num_samples, 2)


# the fused kernel (sampling inside the tile waves) for the shapes it takes; False = always sampler + point splat (the tests
# compare the two bit for bit)
FUSED_SAMPLER = True


def _draw_polylines_fused(heatmaps, hs, ws_, strides, polylines, num_samples, radius, diameter_to_sigma_factor, k_scale,
                          num_points, num_lanes, clear):
    b, l, p, _ = polylines.shape
    dev = polylines.device
    pts = polylines.contiguous()
    counts = None
    if num_points is not None:
        if not (num_points.shape == (b, l) and num_points.device == dev):
            raise RuntimeError("num_points must be of shape [batch, lanes] on the polylines' device")
        _poly._check_sizes(num_points.reshape(-1), p, "num_points")
        counts = num_points.contiguous().view(b * l)
        if counts.dtype not in (torch.int32, torch.int64):
            counts = counts.to(torch.int64)
    if num_lanes is None:
        lanes = _cached(("full", b, l, dev), lambda: torch.full((b,), l, dtype=torch.int32, device=dev))
    else:
        if not (num_lanes.shape == (b,)):
            raise RuntimeError("num_lanes must be of shape [batch]")
        lanes = num_lanes.contiguous()
        if lanes.dtype not in (torch.int32, torch.int64):
            lanes = lanes.to(torch.int64)
    k = len(heatmaps)
    ptrs = (ctypes.c_void_p * k)(*[hm.data_ptr() for hm in heatmaps])
    st = (ctypes.c_float * k)(*strides)
    flags = (_nat.HM_CLEAR if clear else 0) | (_nat.HM_COUNTS_I64 if lanes.dtype == torch.int64 else 0) | \
        (_nat.HM_POINT_COUNTS_I64 if counts is not None and counts.dtype == torch.int64 else 0) | _ops._FORCED_FLAGS
    with _nat.device_guard(dev):
        status = _nat.lib().accv_draw_polylines_multiscale_f32(
            ptrs, hs, ws_, st, k, b, pts.data_ptr(), l, p, counts.data_ptr() if counts is not None else None,
            lanes.data_ptr(), int(num_samples), int(radius), float(diameter_to_sigma_factor), float(k_scale), flags,
            _nat.stream_ptr(dev))
    _nat.check(status, "draw_polylines_multiscale")


def _fused_applies(lib, hs, ws_, k, polylines, num_samples) -> bool:
    b, l = polylines.shape[:2] if polylines.dim() == 4 else (0, 0)
    return bool(FUSED_SAMPLER and b * l > 0 and polylines.dtype == torch.float32 and polylines.dim() == 4 and
                lib.accv_draw_polylines_fused_applicable(hs, ws_, k, b, l, polylines.size(2), num_samples))


# draw_targets_multiscale on a sparse lane set: sampler as a rider + point splat (True) or the one-launch lane raster (False)
TARGETS_PREFER_RIDER = True


class _SamplerJob:
    """The polyline sampler as a rider of the box-map launch (draw_targets_multiscale): where to read the polylines and where to
    write samples and group boxes.  ``run_separately`` is the fall-back when the box maps do not take the one-launch kernel."""

    def __init__(self, polylines, num_points, num_samples, work):
        b, l, p, _ = polylines.shape
        self.polylines, self.num_points_arg, self.num_samples, self.work = polylines, num_points, num_samples, work
        self.points = polylines.contiguous()
        self.num_polylines, self.num_points = b * l, p
        self.counts = None
        self.flags = 0
        if num_points is not None:
            if not (num_points.shape == (b, l) and num_points.device == polylines.device):
                raise RuntimeError("num_points must be of shape [batch, lanes] on the polylines' device")
            _poly._check_sizes(num_points.reshape(-1), p, "num_points")
            self.counts = num_points.contiguous().view(b * l)
            if self.counts.dtype not in (torch.int32, torch.int64):
                self.counts = self.counts.to(torch.int64)
            if self.counts.dtype == torch.int64:
                self.flags = _nat.HM_POINT_COUNTS_I64
        self.samples = torch.empty((b, l * num_samples, 2), dtype=torch.float32, device=polylines.device)

    def run_separately(self):
        self.samples = sample_lanes(self.polylines, self.num_samples, num_points=self.num_points_arg,
                                    group_boxes_ptr=self.work.data_ptr())


def draw_targets_multiscale(heatmaps, centers, bboxes, out_size_factors, lane_heatmaps, polylines: torch.Tensor, num_samples: int,
                            radius: int, lane_out_size_factors=None, diameter_to_sigma_factor: float = 6.0, k_scale: float = 1.0, *,
                            num_points: Optional[torch.Tensor] = None, num_lanes: Optional[torch.Tensor] = None,
                            clear: bool = False) -> None:
    """(extension) Box maps and lane maps of one training step (BASELINE config 3).  Equivalent to::

        draw_heatmap_multiscale(heatmaps, centers, bboxes, out_size_factors, diameter_to_sigma_factor, k_scale, clear=clear)
        draw_polylines_multiscale(lane_heatmaps, polylines, num_samples, radius, lane_out_size_factors or out_size_factors,
                                  diameter_to_sigma_factor, k_scale, num_points=num_points, num_lanes=num_lanes, clear=clear)

    bit for bit, in TWO launches instead of three: the polyline sampler's workgroups ride in the box-map launch, which does not
    depend on them, and the point splat follows.  Needs polylines of at most 64 points and a multiple of 64 samples; other
    shapes run the two calls above.  (Sparse lane sets, which ``draw_polylines_multiscale`` alone rasterises with its one-launch
    kernel, also take the rider here: 32.3 against 33.9 us per step on config 3's maps with one polyline per frame.)"""
    heatmaps, lane_heatmaps = list(heatmaps), list(lane_heatmaps)
    lane_strides = [float(f) for f in (out_size_factors if lane_out_size_factors is None else lane_out_size_factors)]
    shape = polylines.shape if isinstance(polylines, torch.Tensor) else ()
    ok = len(shape) == 4 and shape[3] == 2 and polylines.is_cuda and polylines.dtype == torch.float32 and \
        shape[0] * shape[1] > 0 and shape[2] <= 64 and num_samples % 64 == 0 and 64 <= num_samples <= (1 << 20) and radius >= 0 and \
        len(lane_heatmaps) == len(lane_strides)
    # every map is looked at ONCE here and the arrays are handed on (the two operators below would each walk them again: the
    # python side of a step was 31 us against 38 us of kernels)
    lane_geo = _ops._quick_geometry(lane_heatmaps, shape[0], polylines.device) if ok else None
    box_geo = _ops._quick_geometry(heatmaps, shape[0], polylines.device) if lane_geo is not None else None
    ok = box_geo is not None
    if ok and not TARGETS_PREFER_RIDER:
        ok = not _fused_applies(_nat.lib(), lane_geo[1], lane_geo[2], len(lane_heatmaps), polylines, num_samples)
    if not ok:
        _ops.draw_heatmap_multiscale(heatmaps, centers, bboxes, out_size_factors, diameter_to_sigma_factor, k_scale, clear=clear)
        draw_polylines_multiscale(lane_heatmaps, polylines, num_samples, radius, lane_strides, diameter_to_sigma_factor, k_scale,
                                  num_points=num_points, num_lanes=num_lanes, clear=clear)
        return
    b, l = shape[0], shape[1]
    dev = polylines.device
    nbytes = _nat.lib().accv_draw_points_workspace_bytes(b, l * num_samples)      # (host arithmetic: no device guard needed)
    work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    job = _SamplerJob(polylines, num_points, num_samples, work)
    _ops.draw_heatmap_multiscale(heatmaps, centers, bboxes, out_size_factors, diameter_to_sigma_factor, k_scale, clear=clear,
                                 _sampler_job=job, _geometry=box_geo)
    draw_polylines_multiscale(lane_heatmaps, polylines, num_samples, radius, lane_strides, diameter_to_sigma_factor, k_scale,
                              num_points=num_points, num_lanes=num_lanes, clear=clear, _presampled=(job.samples, work, lane_geo))


def draw_polylines_multiscale(heatmaps, polylines: torch.Tensor, num_samples: int, radius: int, out_size_factors,
                              diameter_to_sigma_factor: float = 6.0, k_scale: float = 1.0, *,
                              num_points: Optional[torch.Tensor] = None, num_lanes: Optional[torch.Tensor] = None,
                              clear: bool = False, _presampled=None) -> None:
    """Lane raster at several strides: equivalent to ``draw_polylines_batched(heatmaps[s], polylines, num_samples, radius,
    out_size_factors[s], ...)`` for every scale, in THREE launches altogether (sampler, group boxes, one splat over the
    tiles of all scales) instead of three per scale.  The splat culls in two levels — 64 consecutive samples share a
    bounding box — so a tile only walks the stretches of lane that can reach it.  Always uses the box-walking small-splat
    arithmetic (meant for radii of a few pixels)."""
    heatmaps = list(heatmaps)
    strides = [float(f) for f in out_size_factors]
    if not (len(heatmaps) == len(strides) and len(heatmaps) >= 1):
        raise RuntimeError("heatmaps and out_size_factors must have the same, non-zero length")
    b, l = polylines.shape[:2] if polylines.dim() == 4 else (0, 0)
    fusable = len(heatmaps) <= 4 and radius >= 0
    geometry = _presampled[2] if _presampled is not None and len(_presampled) > 2 else None   # maps checked by draw_targets_multiscale
    if geometry is None and polylines.dim() == 4:
        geometry = _ops._quick_geometry(heatmaps, b, polylines.device)    # one pass when all is well; else the checks below say why
    for hm in (heatmaps if geometry is None else ()):
        if not (isinstance(hm, torch.Tensor) and hm.is_cuda and hm.is_contiguous() and hm.dtype == torch.float32):
            raise RuntimeError("every heatmap must be a contiguous float32 CUDA tensor")
        if not (hm.dim() == 3 and hm.size(0) == b and hm.device == polylines.device):
            raise RuntimeError("every heatmap must be of shape [batch_size, height, width] on the polylines' device")
        fusable = fusable and hm.size(2) % 4 == 0 and hm.data_ptr() % 16 == 0 and hm.size(1) * hm.size(2) * 4 < (1 << 31)
    if not fusable:
        for hm, f in zip(heatmaps, strides):
            _draw_polylines_via_targets(hm, polylines, num_samples, radius, f, diameter_to_sigma_factor, k_scale,
                                        num_points=num_points, num_lanes=num_lanes, clear=clear)
        return
    n = l * num_samples
    dev = polylines.device
    lib = _nat.lib()
    k = len(heatmaps)
    if geometry is None:
        hs = (ctypes.c_int * k)(*[hm.size(1) for hm in heatmaps])
        ws_ = (ctypes.c_int * k)(*[hm.size(2) for hm in heatmaps])
    else:
        hs, ws_ = geometry[1], geometry[2]
    if _presampled is None and _fused_applies(lib, hs, ws_, k, polylines, num_samples):
        # ONE launch: the tile waves sample the polylines themselves (no sampler launch, no sample buffer)
        _draw_polylines_fused(heatmaps, hs, ws_, strides, polylines, num_samples, radius, diameter_to_sigma_factor, k_scale,
                              num_points, num_lanes, clear)
        return
    if _presampled is not None:   # (draw_targets_multiscale) samples and group boxes were written by the box-map launch
        samples, work = _presampled[0], _presampled[1]
        nbytes = work.numel()
        boxes_by_sampler = True
    else:
        nbytes = lib.accv_draw_points_workspace_bytes(b, n)      # (host arithmetic; the allocation names its device)
        work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # the sampler writes the group boxes itself when the groups of 64 do not straddle lanes (one launch less)
        boxes_by_sampler = num_samples % 64 == 0 and b * l > 0
        samples = sample_lanes(polylines, num_samples, num_points=num_points,
                               group_boxes_ptr=work.data_ptr() if boxes_by_sampler else 0)
    if num_lanes is None:
        sizes = _cached(("full", b, n, dev), lambda: torch.full((b,), n, dtype=torch.int32, device=dev))
    else:
        if not (num_lanes.shape == (b,)):
            raise RuntimeError("num_lanes must be of shape [batch]")
        sizes = num_lanes.clamp(0, l) * num_samples
        if sizes.dtype not in (torch.int32, torch.int64):
            sizes = sizes.to(torch.int64)
    ptrs = geometry[0] if geometry is not None else (ctypes.c_void_p * k)(*[hm.data_ptr() for hm in heatmaps])
    st = (ctypes.c_float * k)(*strides)
    flags = (_nat.HM_CLEAR if clear else 0) | (_nat.HM_COUNTS_I64 if sizes.dtype == torch.int64 else 0) | \
        (_nat.HM_GROUP_BOXES_GIVEN if boxes_by_sampler else 0) | _ops._FORCED_FLAGS
    with _nat.device_guard(dev):
        status = lib.accv_draw_points_multiscale_f32(
            ptrs, hs, ws_, st, k, b, samples.data_ptr(), sizes.data_ptr(), n, int(radius),
            float(diameter_to_sigma_factor), float(k_scale), flags, work.data_ptr(), nbytes, _nat.stream_ptr(dev))
    _nat.check(status, "draw_polylines_multiscale")
