"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path under accv-lab_amd/).

ctypes front end to oracle/h1_splat.c, the CPU restatement of the reference's Gaussian splat
(packages/draw_heatmap/accvlab/draw_heatmap/include/draw_heatmap_cuda_kernel.cuh:36-108 for the index
math, packages/draw_heatmap/tests/_gaussian_drawing.py:215-240 for the values).  Parity: PINNED by
tests/test_oracle_h1.py against tests/golden/h1_*.npz (generated from the reference's python oracle).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, a second or two). Building the checker is not using it."""
    src = os.path.join(_HERE, "h1_splat.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.oracle_draw_heatmap_flat.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, i32p, i32p,
                                                 ctypes.c_int, ctypes.c_double, ctypes.c_float]
        lib.oracle_draw_heatmap_flat.restype = ctypes.c_int
        lib.oracle_draw_heatmap_batched.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    i32p, i32p, i64p, i32p, ctypes.c_int, ctypes.c_double,
                                                    ctypes.c_float, ctypes.c_int, ctypes.c_int]
        lib.oracle_draw_heatmap_batched.restype = ctypes.c_int
        lib.oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _ptr(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def draw_heatmap_flat(heatmaps: np.ndarray, centers, radii, heatmap_idxes, factor=6.0, k=1.0) -> np.ndarray:
    """In-place max-splat into ``heatmaps[P,H,W]`` (float32, C-contiguous). Returns the same array."""
    assert heatmaps.dtype == np.float32 and heatmaps.flags.c_contiguous and heatmaps.ndim == 3
    centers, radii, idx = _i32(centers).reshape(-1, 2), _i32(radii).reshape(-1), _i32(heatmap_idxes).reshape(-1)
    n = radii.shape[0]
    assert centers.shape[0] == n and idx.shape[0] == n
    p, h, w = heatmaps.shape
    _load().oracle_draw_heatmap_flat(_ptr(heatmaps, ctypes.c_float), p, h, w, _ptr(centers, ctypes.c_int32),
                                     _ptr(radii, ctypes.c_int32), _ptr(idx, ctypes.c_int32), n, float(factor),
                                     float(k))
    return heatmaps


def draw_heatmap_batched(heatmap: np.ndarray, centers, radii, counts, labels=None, factor=6.0, k=1.0,
                         clear=False, threads=1) -> np.ndarray:
    """In-place (or clear-then-draw) batched splat; ``heatmap`` is [B,H,W] or, with labels, [B,C,H,W]."""
    assert heatmap.dtype == np.float32 and heatmap.flags.c_contiguous
    centers, radii = _i32(centers), _i32(radii)
    b, nmax = radii.shape
    assert centers.shape == (b, nmax, 2)
    counts = np.ascontiguousarray(counts, dtype=np.int64).reshape(b)
    if labels is None:
        assert heatmap.ndim == 3 and heatmap.shape[0] == b
        c, (h, w) = 0, heatmap.shape[1:]
        lab_p = None
    else:
        labels = _i32(labels)
        assert heatmap.ndim == 4 and heatmap.shape[0] == b and labels.shape == (b, nmax)
        c, h, w = heatmap.shape[1:]
        lab_p = _ptr(labels, ctypes.c_int32)
    _load().oracle_draw_heatmap_batched(_ptr(heatmap, ctypes.c_float), b, c, h, w, _ptr(centers, ctypes.c_int32),
                                        _ptr(radii, ctypes.c_int32), _ptr(counts, ctypes.c_int64), lab_p, nmax,
                                        float(factor), float(k), int(bool(clear)), int(threads))
    return heatmap


def max_threads() -> int:
    return int(_load().oracle_max_threads())
