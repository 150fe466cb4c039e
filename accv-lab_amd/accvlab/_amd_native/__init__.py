"""ctypes binding of libaccv_hip.so (the gfx950 C-ABI declared in include/accv_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised — the product path never routes through a CPU or eager-PyTorch substitute.
Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C accv-lab_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACCV_HIP_LIB points experiments (scripts/h1_variants.py) at another build, e.g. the A/B build `make tune`
# ACCV_NO_HOST_FASTPATH=1: the operators keep to their python formulation (the C++ host fast paths of draw_heatmap_batched,
# the ragged gather / scatter and the lane sampler decline everything) — used to run the test-suite over both
NO_HOST_FASTPATH = os.environ.get("ACCV_NO_HOST_FASTPATH", "") not in ("", "0")
LIB_PATH = os.environ.get("ACCV_HIP_LIB") or os.path.join(_HERE, "libaccv_hip.so")

OK = 0
HM_CLEAR = 1
HM_COUNTS_I64 = 2
HM_SMALL_RADII = 4
HM_WRITE_THROUGH = 8
HM_GROUP_BOXES_GIVEN = 16
HM_TILE_ROWS_16 = 32
HM_TILE_ROWS_8 = 64
HM_CALLER_SCALE_ORDER = 256
HM_PLAIN_STORES = 128
HM_POINT_COUNTS_I64 = 512
MP_IDX_I64 = 1
MP_COUNTS_I64 = 2
MP_LABELS_I64 = 4

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_u = ctypes.c_uint
_sz = ctypes.c_size_t
_i64 = ctypes.c_int64
_ll = ctypes.c_longlong
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must list every symbol include/accv_hip.h declares (tests check this)
SIGNATURES = {
    "accv_last_error": (ctypes.c_char_p, []),
    "accv_version": (_i, []),
    "accv_draw_heatmap_last_dispatch": (ctypes.c_char_p, []),
    "accv_draw_heatmap_time_next_launch": (_i, [_vp, _vp]),
    "accv_draw_heatmap_flat_workspace_bytes": (_sz, [_i, _i]),
    "accv_draw_heatmap_flat_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _f, _f, _u, _vp, _sz, _vp]),
    "accv_draw_heatmap_batched_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _f, _u, _vp]),
    "accv_fill_f32": (_i, [_vp, _sz, _f, _vp]),
    "accv_draw_heatmap_multiscale_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _f, _f, _u, _vp]),
    "accv_draw_heatmap_multiscale_sample_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _f, _f, _u,
                                                      _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "accv_draw_points_workspace_bytes": (_sz, [_i, _i]),
    "accv_draw_points_multiscale_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _f, _f, _u, _vp, _sz, _vp]),
    "accv_draw_polylines_fused_applicable": (_i, [_vp, _vp, _i, _i, _i, _i, _i]),
    "accv_draw_polylines_multiscale_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _f, _f, _u, _vp]),
    "accv_heatmap_targets_from_boxes_f32": (_i, [_vp, _vp, _ll, _f, _vp, _vp, _vp]),
    "accv_heatmap_targets_from_points_f32": (_i, [_vp, _ll, _f, _i, _vp, _vp, _vp]),
    # H2 ragged kernels
    "accv_ragged_gather": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _i, _i, _vp, _vp]),
    "accv_ragged_gather_fill": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _u64, _i, _i, _i, _vp, _vp]),
    "accv_ragged_scatter": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _i, _i, _vp, _vp]),
    "accv_ragged_map_pairs": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _i, _vp, _vp]),
    "accv_ragged_insert_const": (_i, [_vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _u64, _i, _i, _i, _vp, _vp]),
    "accv_ragged_pad_fill": (_i, [_vp, _vp, _ll, _ll, _ll, _u64, _i, _i, _vp]),
    "accv_ragged_accumulate": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _i, _i, _vp, _vp]),
    "accv_ragged_mask_to_indices": (_i, [_vp, _vp, _i, _ll, _ll, _vp, _vp, _vp]),
    "accv_ragged_mask_to_indices_workspace_bytes": (_sz, [_ll, _ll]),
    "accv_ragged_mask_to_indices_ws": (_i, [_vp, _vp, _i, _ll, _ll, _vp, _vp, _vp, _sz, _vp]),
    "accv_ragged_pack": (_i, [_vp, _vp, _vp, _vp, _ll, _ll, _ll, _i, _vp]),
    "accv_matched_pair_reduce_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _f, _i, _i, _vp, _vp]),
    "accv_matched_pair_reduce_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _f, _i, _i,
                                              _vp, _vp, _vp, _vp]),
    "accv_matched_pair_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _i, _f, _f, _u, _vp, _vp]),
    "accv_matched_pair_reduce_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _ll, _ll, _ll, _ll, _i, _i, _f, _f, _u,
                                          _vp, _vp, _vp, _vp]),
    # H3 multi-tensor copier
    "accv_mtc_plan": (_i, [_ll, _vp, _vp, _vp, _ll, _ll, _vp, _vp, _vp, _vp]),
    "accv_pinned_acquire": (_vp, [_sz]),
    "accv_pinned_release": (None, [_vp]),
    "accv_pinned_trim": (None, []),
    "accv_pinned_total_bytes": (_sz, []),
    "accv_mtc_worker_count": (_i, []),
    "accv_mtc_pack_host": (_i, [_ll, _vp, _vp, _vp, _vp, _ll]),
    "accv_mtc_stage_h2d": (_i, [_ll, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _i]),
    "accv_mtc_stage_h2d_async": (_i, [_ll, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "accv_mtc_async_wait": (_i, [_ll]),
    "accv_mtc_async_poll": (_i, [_ll]),
    "accv_mtc_shutdown": (None, []),
    "accv_mtc_async_tickets_held": (_ll, []),
    "accv_mtc_coalesce": (_i, [_vp, _ll, _vp, _i, _vp]),
    "accv_memcpy_async": (_i, [_vp, _vp, _sz, _i, _vp]),
    # lane_helpers
    "accv_polyline_scratch_bytes": (_sz, [_ll, _i, _i]),
    "accv_polyline_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "accv_polyline_sample_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i]),
    "accv_polyline_sample_boxes": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
}
# not in the public header and not in the shipped library: the knob setter of the A/B build (make -C csrc tune)
_PRIVATE = {
    "accv_tune_set": (_i, [ctypes.c_char_p, _i]),
}

_lib = None
_handle = None

try:  # METH_FASTCALL trampoline (csrc_host/fastcall.cpp): same exported functions, ~0.3 us per call instead of ~4 us
    from . import _fastcall
except ImportError:  # pragma: no cover - the ctypes path is complete on its own
    _fastcall = None
if os.environ.get("ACCV_NO_FASTCALL") == "1":
    _fastcall = None

_INT_CLASS = (_vp, _i, _u, _sz, _i64, _ll, _u64)
# entry points that BLOCK (wait for a native job, run a long host memcpy): they stay on ctypes, which drops the
# interpreter lock for the duration of the call — the trampoline keeps it
_BLOCKING = {"accv_mtc_async_wait", "accv_mtc_stage_h2d", "accv_mtc_pack_host", "accv_polyline_sample_host"}


def _fast_entry(fn, res, args):
    """A callable with the C argument order that goes through the trampoline, or None when the signature is not
    eligible (non-int result, string arguments, a float count other than 0 or 2)."""
    if _fastcall is None or res is not _i or len(args) > 22:
        return None
    floats = [k for k, a in enumerate(args) if a is _f]
    if any(a is not _f and a not in _INT_CLASS for a in args):
        return None
    addr = ctypes.cast(fn, ctypes.c_void_p).value
    if not floats and len(args) <= 20:
        import functools
        return functools.partial(_fastcall.call_ints, addr)
    if len(floats) == 2 and floats[1] == floats[0] + 1 and len(args) - 2 <= 20:
        f0, call = floats[0], _fastcall.call_f2
        return lambda *a: call(addr, a[f0], a[f0 + 1], *a[:f0], *a[f0 + 2:])
    return None


class _Lib:
    """Attribute access by exported name; hot entry points resolve to the trampoline, everything else to ctypes."""

    def __init__(self, handle):
        self._ctypes = handle

    def __getattr__(self, name):
        fn = getattr(self._ctypes, name)
        sig = SIGNATURES.get(name) or _PRIVATE.get(name)
        fast = _fast_entry(fn, *sig) if sig and name not in _BLOCKING else None
        fn = fast or fn
        setattr(self, name, fn)       # cached: __getattr__ is not consulted again
        return fn


class AccvNativeError(RuntimeError):
    """Raised when a libaccv_hip.so entry point reports a failure (mirrors the reference's TORCH_CHECK ->
    RuntimeError behaviour)."""


def lib() -> "_Lib":
    global _lib, _handle
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C accv-lab_amd/csrc` or __graft_entry__.build()). There is no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for table in (SIGNATURES, _PRIVATE):
            for name, (res, args) in table.items():
                if table is _PRIVATE and not hasattr(handle, name):
                    continue
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
        _handle = handle
        _lib = _Lib(handle)
    return _lib


def ctypes_lib() -> ctypes.CDLL:
    """The plain ctypes handle (symbol-table checks, tests)."""
    lib()
    return _handle


def check(status: int, what: str = "") -> None:
    if status != OK:
        msg = lib().accv_last_error()
        raise AccvNativeError(f"{what}: {msg.decode() if msg else 'error'} (status {status})")


def tune_set(key: str, value: int) -> None:
    """Kernel-variant knobs exist only in the A/B build (``make -C accv-lab_amd/csrc tune`` + ``ACCV_HIP_LIB``)."""
    handle = ctypes_lib()
    if not hasattr(handle, "accv_tune_set"):
        raise AccvNativeError("this libaccv_hip.so has no tuning knobs (shipped build); use the public flags "
                              "(tile_rows=, write_through=, small_radii=) or load the A/B build via ACCV_HIP_LIB")
    check(handle.accv_tune_set(key.encode(), int(value)), "accv_tune_set")


def last_dispatch() -> str:
    """Kernel instantiation + launch geometry of the last draw_heatmap call on this thread."""
    s = ctypes_lib().accv_draw_heatmap_last_dispatch()
    return s.decode() if s else ""


def stream_ptr(device) -> int:
    """hipStream_t of torch's CURRENT stream on `device` (the reference launches on
    at::cuda::getCurrentCUDAStream(), draw_heatmap_cuda.cu:65)."""
    import torch

    idx = device.index
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the accessor torch's own generated code uses
    if raw is None:  # pragma: no cover - other torch builds
        return torch.cuda.current_stream(device).cuda_stream
    return raw(torch.cuda.current_device() if idx is None else idx)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """Context manager that makes `device` current for a launch (the reference's at::DeviceGuard,
    draw_heatmap_cuda.cu:64) — free when it already is, which is the case in one-process-per-GPU training."""
    import torch

    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(idx)
