"""Host side of the Gaussian heat-map rasteriser: argument validation + one C-ABI call per op.

Mirrors the checks of the reference launchers (packages/draw_heatmap/accvlab/draw_heatmap/csrc/
draw_heatmap_cuda.cu:62-89, 91-124, 126-165) and of the python wrapper
(funtions/draw_heatmap_batched.py:27-84): tensor-property violations raise ``RuntimeError``,
RaggedBatch shape mismatches raise ``AssertionError``.  All work is enqueued on torch's current stream of
the heat-map's device; nothing synchronises.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _amd_native as _nat

try:  # C++ fast path of the plain draw_heatmap_batched call (built by `make -C accv-lab_amd/csrc_host`); complete without it
    from . import _dh_host as _dh
except ImportError:  # pragma: no cover
    _dh = None
_native_bound = False


def _native():
    """_dh_host with the C-ABI addresses bound (same library handle as the ctypes binding), or None."""
    global _native_bound
    if _dh is None or _nat.NO_HOST_FASTPATH:
        return None
    if not _native_bound:
        h = _nat.ctypes_lib()
        addr = lambda name: ctypes.cast(getattr(h, name), ctypes.c_void_p).value  # noqa: E731
        _dh.bind_native(addr("accv_draw_heatmap_batched_f32"), addr("accv_last_error"))
        _native_bound = True
    return _dh


# OR-ed into the flags of every draw_heatmap / draw_heatmap_batched call: lets the GPU test-suite run each test against
# every kernel instantiation the public hints can select (tests/test_draw_heatmap_gpu.py); 0 in production
_FORCED_FLAGS = 0


def _hint_flags(clear: bool, small_radii: bool, write_through: bool, tile_rows) -> int:
    if tile_rows not in (None, 8, 16):
        raise RuntimeError("tile_rows must be None, 8 or 16")
    return (_nat.HM_CLEAR if clear else 0) | (_nat.HM_SMALL_RADII if small_radii else 0) | \
        (0 if write_through is None else _nat.HM_WRITE_THROUGH if write_through else _nat.HM_PLAIN_STORES) | \
        (_nat.HM_TILE_ROWS_16 if tile_rows == 16 else _nat.HM_TILE_ROWS_8 if tile_rows == 8 else 0) | _FORCED_FLAGS


def _quick_geometry(maps, batch: int, device):
    """``(ptrs, heights, widths)`` as ctypes arrays when every map is a contiguous float32 ``[batch, H, W]`` CUDA tensor on
    ``device`` that the one-launch multi-scale kernels take (width a multiple of 4, 16-byte aligned, planes below 2 GiB), else
    ``None`` — one pass over the maps for callers that hand the arrays on (draw_targets_multiscale)."""
    k = len(maps)
    if not 1 <= k <= 4:
        return None
    ptrs, hs, ws = [], [], []
    for hm in maps:
        if not (isinstance(hm, torch.Tensor) and hm.is_cuda and hm.dtype == torch.float32 and hm.is_contiguous() and
                hm.device == device):
            return None
        shape = hm.shape
        if len(shape) != 3 or shape[0] != batch:
            return None
        ptr = hm.data_ptr()
        if shape[2] % 4 or ptr % 16 or shape[1] * shape[2] * 4 >= (1 << 31):
            return None
        ptrs.append(ptr)
        hs.append(shape[1])
        ws.append(shape[2])
    return (ctypes.c_void_p * k)(*ptrs), (ctypes.c_int * k)(*hs), (ctypes.c_int * k)(*ws)


def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _check_input(t: torch.Tensor, name: str) -> None:
    if not (isinstance(t, torch.Tensor)):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if not (t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not (t.is_contiguous()):
        raise RuntimeError(f"{name} must be contiguous")


def _check_dtype(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if not (t.dtype == dtype):
        raise RuntimeError(f"{name}: expected scalar type {dtype} but found {t.dtype}")


def _same_device(ref: torch.Tensor, *others) -> None:
    for name, t in others:
        if not (t.device == ref.device):
            raise RuntimeError(f"{name} must be on the same device as the heatmap ({ref.device})")


_FLAT_DIRECT_MAX_OBJECTS = 2048
_FLAT_DIRECT_MAX_ROUNDS = 200_000     # tiles x cull rounds of 64 objects: ~8 M wave instructions, a few microseconds of the chip
_count_cache = {}


def _flat_direct(planes: int, height: int, width: int, n: int) -> bool:
    if not (0 < n <= _FLAT_DIRECT_MAX_OBJECTS and 0 < planes <= 65535 and height > 0 and width > 0):
        return False
    tiles = planes * ((width + 127) // 128) * ((height + 15) // 16)
    return tiles * ((n + 63) // 64) <= _FLAT_DIRECT_MAX_ROUNDS


def _one_count(n: int, device) -> torch.Tensor:
    """int32 [1] = n on `device`, shared by later calls on any stream (created once: its stream is drained before it is
    published; nothing is cached while a stream is being captured)."""
    key = (n, device)
    t = _count_cache.get(key)
    if t is None:
        t = torch.full((1,), n, dtype=torch.int32, device=device)
        if torch.cuda.is_current_stream_capturing():
            return t
        torch.cuda.current_stream(device).synchronize()
        if len(_count_cache) > 256:
            _count_cache.clear()
        _count_cache[key] = t
    return t


def draw_heatmap(
    heatmaps: torch.Tensor,
    centers: torch.Tensor,
    radii: torch.Tensor,
    heatmap_idxes: torch.Tensor,
    diameter_to_sigma_factor: float = 6.0,
    k_scale: float = 1.0,
    *,
    clear: bool = False,
    small_radii: bool = False,
    write_through: Optional[bool] = None,
    tile_rows: Optional[int] = None,
) -> None:
    """Draw N Gaussians into ``heatmaps[P,H,W]`` (fp32, in place, element-wise max).

    Args:
        heatmaps: float32 ``[num_heatmaps, height, width]``, modified in place.
        centers: int32 ``[N, 2]`` as (x, y).
        radii: int32 ``[N]``.
        heatmap_idxes: int32 ``[N]`` — plane each object is drawn into.
        diameter_to_sigma_factor: sigma = (2r+1) / factor.
        k_scale: peak value.
        clear: (extension) overwrite the map with max(0, splats) instead of max-ing into its content.
        small_radii: (extension) performance hint — the radii are a few pixels (key points, lane samples; boxes up to
            ~15x15): use the kernel that walks each object's box instead of updating whole tiles.  Same results.
        write_through: (extension) performance hint — True: write-through non-temporal stores everywhere (pays a few per
            cent for dense launches that rewrite hundreds of MB, costs up to 27 % for sparse in-place ones); False: plain
            stores everywhere; None (default): plain for fused-clear launches, and for in-place launches the kernel
            decides per plane from the density of its objects.  Same results.

    Reference: draw_heatmap.cpp:132-134 -> draw_heatmap_launcher (draw_heatmap_cuda.cu:62-89).
    """
    _check_input(heatmaps, "heatmap")
    _check_input(centers, "centers")
    _check_input(radii, "radii")
    _check_input(heatmap_idxes, "heatmap_idxes")
    if not (centers.size(0) == radii.size(0)):
        raise RuntimeError("centers and radii must have the same size at dim0")
    if not (centers.size(0) == heatmap_idxes.size(0)):
        raise RuntimeError("centers and heatmap_idxes must have the same size at dim0")
    if not (heatmaps.dim() == 3):
        raise RuntimeError("heatmap must be of shape [num_heatmaps, height, width]")
    if not (centers.dim() == 2 and centers.size(1) == 2):
        raise RuntimeError("centers must be of shape [num_targets, 2]")
    if not (radii.dim() == 1 and heatmap_idxes.dim() == 1):
        raise RuntimeError("radii and heatmap_idxes must be of shape [num_targets]")
    _check_dtype(heatmaps, torch.float32, "heatmap")
    _check_dtype(centers, torch.int32, "centers")
    _check_dtype(radii, torch.int32, "radii")
    _check_dtype(heatmap_idxes, torch.int32, "heatmap_idxes")
    _same_device(heatmaps, ("centers", centers), ("radii", radii), ("heatmap_idxes", heatmap_idxes))

    lib = _nat.lib()
    planes, height, width = heatmaps.shape
    n = centers.size(0)
    if _flat_direct(planes, height, width, n):
        # few objects: ONE launch.  The flat input is the class-wise batched call with a single sample whose "classes" are
        # the planes (plane = label; labels outside [0, P) match no plane, i.e. such objects are ignored, as in the flat
        # entry point): every tile tests all N objects instead of its plane's share, which costs less than the binning
        # launch while N is small.  Same values (max is order independent).
        with _nat.device_guard(heatmaps.device):
            count = _one_count(n, heatmaps.device)
            status = lib.accv_draw_heatmap_batched_f32(
                heatmaps.data_ptr(), 1, planes, height, width, centers.data_ptr(), radii.data_ptr(), count.data_ptr(),
                heatmap_idxes.data_ptr(), n, float(diameter_to_sigma_factor), float(k_scale),
                _hint_flags(clear, small_radii, write_through, tile_rows), _nat.stream_ptr(heatmaps.device))
        _nat.check(status, "draw_heatmap")
        return
    with _nat.device_guard(heatmaps.device):
        ws_bytes = lib.accv_draw_heatmap_flat_workspace_bytes(planes, n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=heatmaps.device)
        status = lib.accv_draw_heatmap_flat_f32(
            heatmaps.data_ptr(), planes, height, width, centers.data_ptr(), radii.data_ptr(),
            heatmap_idxes.data_ptr(), n, float(diameter_to_sigma_factor), float(k_scale),
            _hint_flags(clear, small_radii, write_through, tile_rows), ws.data_ptr(), ws_bytes,
            _nat.stream_ptr(heatmaps.device))
        # the workspace is only used by kernels already enqueued on the current stream; the caching
        # allocator re-issues it stream-ordered, so dropping the reference here is safe
    _nat.check(status, "draw_heatmap")


def draw_heatmap_batched(
    heatmap: torch.Tensor,
    centers,
    radii,
    diameter_to_sigma_factor: float = 6.0,
    k_scale: float = 1.0,
    labels=None,
    *,
    clear: bool = False,
    small_radii: bool = False,
    write_through: Optional[bool] = None,
    tile_rows: Optional[int] = None,
) -> None:
    """Draw a ragged batch of Gaussians (in place, element-wise max).

    Args:
        heatmap: float32 ``[B, H, W]`` when ``labels`` is None, else ``[B, C, H, W]``.
        centers: RaggedBatch (anything with ``.tensor`` and ``.sample_sizes``) int32 ``[B, Nmax, 2]`` (x, y).
        radii: RaggedBatch int32 ``[B, Nmax]``.
        diameter_to_sigma_factor: sigma = (2r+1) / factor.
        k_scale: peak value.
        labels: optional RaggedBatch int32 ``[B, Nmax]`` of class indices -> one plane per class.
        clear: (extension) fused zero-fill + draw in one write-only pass.
        small_radii: (extension) performance hint for point-like objects, see :func:`draw_heatmap`.
        write_through: (extension) store-policy hint for dense launches over huge maps, see :func:`draw_heatmap`.
        tile_rows: (extension) tile-height hint (8 or 16), see :func:`draw_heatmap`.

    Only ``centers.sample_sizes`` decides how many leading objects of a sample are drawn; padded slots are
    never touched.  Reference: funtions/draw_heatmap_batched.py:27-84 -> draw_heatmap_batched_launcher /
    draw_heatmap_batched_classwise_launcher (draw_heatmap_cuda.cu:91-165).
    """
    centers_t = centers.tensor
    radii_t = radii.tensor
    assert centers_t.shape[0] == radii_t.shape[0], "centers and radii must have the same size batch size"
    assert centers_t.shape[1] == radii_t.shape[1], "centers and radii must have the same maximum number of objects"
    counts = centers.sample_sizes
    labels_t: Optional[torch.Tensor] = None
    if labels is not None:
        labels_t = labels.tensor
        assert centers_t.shape[0] == labels_t.shape[0], "centers and labels must have the same size batch size"
        assert centers_t.shape[1] == labels_t.shape[1], \
            "centers and labels must have the same maximum number of objects"

    # plain call (CUDA, contiguous, expected dtypes and extents): the launcher's checks and the launch in C++; anything else
    # is declined there and takes the checks below, which raise the reference's errors
    native = _native()
    if native is not None and isinstance(heatmap, torch.Tensor) and isinstance(counts, torch.Tensor) and \
            native.draw_batched(heatmap, centers_t, radii_t, counts, labels_t, float(diameter_to_sigma_factor), float(k_scale),
                                _hint_flags(clear, small_radii, write_through, tile_rows)):
        return

    # the reference casts to int32 with an extra kernel on every call (draw_heatmap_batched.py:63);
    # the C-ABI reads int32 or int64 counts directly
    if counts.dtype not in (torch.int32, torch.int64):
        counts = counts.to(torch.int64)

    _check_input(heatmap, "heatmap")
    _check_input(centers_t, "centers")
    _check_input(radii_t, "radii")
    _check_input(counts, "nums_targets")
    batch = heatmap.size(0)
    n_max = radii_t.size(1) if radii_t.dim() >= 2 else -1
    _require(batch == radii_t.size(0) and batch == centers_t.size(0) and batch == counts.size(0),
             "batch_size (dim 0) need to be the same for all inputs")
    if not (centers_t.dim() == 3 and centers_t.size(2) == 2):
        raise RuntimeError("centers must be of shape [batch_size, num_targets, 2]")
    if not (radii_t.dim() == 2):
        raise RuntimeError("radii must be of shape [batch_size, num_targets]")
    if not (n_max == centers_t.size(1)):
        raise RuntimeError("maximum number of targets (dim 1) need to be the same centers and radii")
    if not (counts.dim() == 1):
        raise RuntimeError("nums_targets must be of shape [batch_size]")
    _check_dtype(heatmap, torch.float32, "heatmap")
    _check_dtype(centers_t, torch.int32, "centers")
    _check_dtype(radii_t, torch.int32, "radii")
    others = [("centers", centers_t), ("radii", radii_t), ("nums_targets", counts)]
    if labels_t is None:
        if not (heatmap.dim() == 3):
            raise RuntimeError("heatmap must be of shape [batch_size, height, width]")
        num_classes, (height, width) = 0, heatmap.shape[1:]
        labels_ptr = None
    else:
        _check_input(labels_t, "labels")
        if not (heatmap.dim() == 4):
            raise RuntimeError("heatmap must be of shape [batch_size, max_num_classes, height, width]")
        if not (labels_t.dim() == 2):
            raise RuntimeError("labels must be of shape [batch_size, radii.size(1)]")
        _require(labels_t.size(0) == batch and labels_t.size(1) == n_max,
                 "labels shape must be [batch_size, radii.size(1)]")
        _check_dtype(labels_t, torch.int32, "labels")
        num_classes, height, width = heatmap.shape[1:]
        if not (num_classes > 0 or heatmap.numel() == 0):
            raise RuntimeError("class-wise heatmap needs at least one class plane")
        labels_ptr = labels_t.data_ptr()
        others.append(("labels", labels_t))
    _same_device(heatmap, *others)
    if labels_t is not None and num_classes == 0:
        return

    flags = _hint_flags(clear, small_radii, write_through, tile_rows) | \
        (_nat.HM_COUNTS_I64 if counts.dtype == torch.int64 else 0)
    with _nat.device_guard(heatmap.device):
        status = _nat.lib().accv_draw_heatmap_batched_f32(
            heatmap.data_ptr(), batch, num_classes, height, width, centers_t.data_ptr(), radii_t.data_ptr(),
            counts.data_ptr(), labels_ptr, n_max, float(diameter_to_sigma_factor), float(k_scale), flags,
            _nat.stream_ptr(heatmap.device))
    _nat.check(status, "draw_heatmap_batched")


def get_centers_and_radii(centers, bboxes, out_size_factor: float):
    """Target-prep front end: float centres ``[..., 2]`` (x, y) and boxes ``[..., 4]`` (x0, y0, x1, y1) in source-image
    pixels -> ``(centers_int32 [..., 2], radii_int32 [...])`` at the heat-map stride ``out_size_factor``:
    ``r = max(1, ceil(min(c - top_left, bottom_right - c) / stride))``, ``c = int(c / stride)``.

    Same semantics as the helper the reference's tests and examples use in front of ``draw_heatmap``
    (packages/draw_heatmap/tests/_test_helpers.py:20-28).  Accepts tensors or RaggedBatch-like objects (then the
    result shares their sample sizes).  CUDA tensors run one fused kernel; CPU tensors use torch.
    """
    ragged = hasattr(centers, "tensor") and hasattr(centers, "create_with_sample_sizes_like_self")
    c_t = centers.tensor if ragged else centers
    b_t = bboxes.tensor if hasattr(bboxes, "tensor") else bboxes
    _require(c_t.shape[-1] == 2 and b_t.shape[-1] == 4 and c_t.shape[:-1] == b_t.shape[:-1],
             "centers must be [..., 2] and bboxes [..., 4] with equal leading dimensions")
    if c_t.is_cuda:
        if not (b_t.device == c_t.device):
            raise RuntimeError("centers and bboxes must be on the same device")
        c32 = c_t.to(torch.float32).contiguous()
        b32 = b_t.to(torch.float32).contiguous()
        out_c = torch.empty(c32.shape, dtype=torch.int32, device=c32.device)
        out_r = torch.empty(c32.shape[:-1], dtype=torch.int32, device=c32.device)
        with _nat.device_guard(c32.device):
            _nat.check(_nat.lib().accv_heatmap_targets_from_boxes_f32(
                c32.data_ptr(), b32.data_ptr(), out_r.numel(), float(out_size_factor), out_c.data_ptr(),
                out_r.data_ptr(), _nat.stream_ptr(c32.device)), "get_centers_and_radii")
    else:
        d = torch.cat([c_t - b_t[..., :2], b_t[..., 2:] - c_t], dim=-1)
        out_r = torch.ceil(torch.min(d, dim=-1)[0] / out_size_factor).to(torch.int32).clamp_(min=1)
        out_c = (c_t / out_size_factor).to(torch.int32)
    if ragged:
        nu = centers.non_uniform_dim
        return centers.create_with_sample_sizes_like_self(out_c, nu), centers.create_with_sample_sizes_like_self(out_r, nu)
    return out_c, out_r


def draw_heatmap_multiscale(heatmaps, centers, bboxes, out_size_factors, diameter_to_sigma_factor: float = 6.0,
                            k_scale: float = 1.0, *, clear: bool = False, _sampler_job=None, _geometry=None) -> None:
    """(extension) Rasterise one batch of objects at several strides.  Equivalent to, for every scale ``s``::

        c, r = get_centers_and_radii(centers, bboxes, out_size_factors[s])
        draw_heatmap_batched(heatmaps[s], c, r, diameter_to_sigma_factor, k_scale, clear=clear)

    but up to four scales run as ONE kernel launch that converts the float centres / boxes inside its culling step
    (no intermediate tensors): small target maps are launch bound, not bandwidth bound.

    Args:
        heatmaps: sequence of float32 ``[B, H_s, W_s]`` tensors (modified in place).
        centers: RaggedBatch float32 ``[B, Nmax, 2]`` (x, y) in source-image pixels.
        bboxes: RaggedBatch (or tensor) float32 ``[B, Nmax, 4]`` (x0, y0, x1, y1) in source-image pixels.
        out_size_factors: one stride per heat-map.
    """
    heatmaps = list(heatmaps)
    strides = [float(f) for f in out_size_factors]
    if not (len(heatmaps) == len(strides) and len(heatmaps) >= 1):
        raise RuntimeError("heatmaps and out_size_factors must have the same, non-zero length")
    c_t = centers.tensor
    b_t = bboxes.tensor if hasattr(bboxes, "tensor") else bboxes
    counts = centers.sample_sizes
    if counts.dtype not in (torch.int32, torch.int64):
        counts = counts.to(torch.int64)
    _check_input(c_t, "centers")
    _check_input(b_t, "bboxes")
    _check_input(counts, "nums_targets")
    if not (c_t.dim() == 3 and c_t.size(2) == 2):
        raise RuntimeError("centers must be of shape [batch_size, num_targets, 2]")
    if not (b_t.dim() == 3 and b_t.size(2) == 4 and b_t.shape[:2] == c_t.shape[:2]):
        raise RuntimeError("bboxes must be of shape [batch_size, num_targets, 4]")
    _check_dtype(c_t, torch.float32, "centers")
    _check_dtype(b_t, torch.float32, "bboxes")
    batch, n_max = c_t.shape[:2]
    if not (counts.dim() == 1 and counts.size(0) == batch):
        raise RuntimeError("nums_targets must be of shape [batch_size]")
    fusable = len(heatmaps) <= 4
    if _geometry is None:
        _geometry = _quick_geometry(heatmaps, batch, c_t.device)    # one pass when all is well; else the checks below say why
    if _geometry is not None:
        _same_device(c_t, ("bboxes", b_t), ("nums_targets", counts))     # (the maps are on the centres' device)
    for hm in (heatmaps if _geometry is None else ()):     # (_geometry given: draw_targets_multiscale has checked the maps already)
        _check_input(hm, "heatmap")
        _check_dtype(hm, torch.float32, "heatmap")
        if not (hm.dim() == 3 and hm.size(0) == batch):
            raise RuntimeError("every heatmap must be of shape [batch_size, height, width]")
        _same_device(hm, ("centers", c_t), ("bboxes", b_t), ("nums_targets", counts))
        fusable = fusable and hm.size(2) % 4 == 0 and hm.data_ptr() % 16 == 0 and hm.size(1) * hm.size(2) * 4 < (1 << 31)
    if not fusable:   # odd widths / more than four scales: the per-scale operators
        for hm, f in zip(heatmaps, strides):
            ci, ri = get_centers_and_radii(centers, bboxes, f)
            draw_heatmap_batched(hm, ci, ri, diameter_to_sigma_factor, k_scale, clear=clear)
        if _sampler_job is not None:
            _sampler_job.run_separately()
        return
    n = len(heatmaps)
    if _geometry is None:
        ptrs = (ctypes.c_void_p * n)(*[hm.data_ptr() for hm in heatmaps])
        hs = (ctypes.c_int * n)(*[hm.size(1) for hm in heatmaps])
        ws = (ctypes.c_int * n)(*[hm.size(2) for hm in heatmaps])
    else:
        ptrs, hs, ws = _geometry
    st = (ctypes.c_float * n)(*strides)
    flags = (_nat.HM_CLEAR if clear else 0) | (_nat.HM_COUNTS_I64 if counts.dtype == torch.int64 else 0) | _FORCED_FLAGS
    dev = heatmaps[0].device
    with _nat.device_guard(dev):
        if _sampler_job is None:
            status = _nat.lib().accv_draw_heatmap_multiscale_f32(
                ptrs, hs, ws, st, n, batch, c_t.data_ptr(), b_t.data_ptr(), counts.data_ptr(), n_max,
                float(diameter_to_sigma_factor), float(k_scale), flags, _nat.stream_ptr(dev))
        else:   # (draw_targets_multiscale) the polyline sampler rides in the same launch
            j = _sampler_job
            status = _nat.lib().accv_draw_heatmap_multiscale_sample_f32(
                ptrs, hs, ws, st, n, batch, c_t.data_ptr(), b_t.data_ptr(), counts.data_ptr(), n_max,
                float(diameter_to_sigma_factor), float(k_scale), flags | j.flags, j.points.data_ptr(), j.num_polylines, j.num_points,
                j.counts.data_ptr() if j.counts is not None else None, j.num_samples, j.samples.data_ptr(), j.work.data_ptr(),
                _nat.stream_ptr(dev))
    _nat.check(status, "draw_heatmap_multiscale")
