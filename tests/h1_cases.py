"""Loaders for the committed H1 golden vectors (tests/golden/h1_*.npz, made by make_h1_golden.py)."""
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def g3_cases():
    z = load("h1_g3.npz")
    names = sorted({k.split("__")[0] for k in z.files})
    for n in names:
        H, W, k, factor, base = z[f"{n}__meta"].tolist()
        yield n, int(H), int(W), z[f"{n}__centers"], z[f"{n}__radii"], float(k), float(factor), float(base), \
            z[f"{n}__expected"]


def g4_frames():
    z = load("h1_g4.npz")
    H, W, T = int(z["H"]), int(z["W"]), int(z["tile"])
    for n in ("A0", "A1", "B0"):
        yield n, H, W, T, z[f"{n}__centers"], z[f"{n}__radii"], z[f"{n}__crc"], z[f"{n}__sums"], \
            z[f"{n}__picks"], z[f"{n}__tiles"]


def tile_crc_and_sums(a, T):
    H, W = a.shape
    ty, tx = (H + T - 1) // T, W // T
    crc = np.zeros((ty, tx), dtype=np.uint32)
    sums = np.zeros((ty, tx), dtype=np.float64)
    for i in range(ty):
        for j in range(tx):
            t = np.ascontiguousarray(a[i * T:(i + 1) * T, j * T:(j + 1) * T])
            crc[i, j] = zlib.crc32(t.tobytes())
            sums[i, j] = t.astype(np.float64).sum()
    return crc, sums
