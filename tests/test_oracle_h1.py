"""Pins the CPU oracle (oracle/h1_splat.c) against golden vectors produced by the reference's own python
oracle (packages/draw_heatmap/tests/_gaussian_drawing.py:215-240).  CPU only."""
import numpy as np
import pytest

import h1_cases
from oracle import h1 as oracle


def _flat_from_padded(centers, radii, sizes, labels=None, C=0):
    cs, rs, idx = [], [], []
    for s, n in enumerate(sizes.tolist()):
        cs.append(centers[s, :n])
        rs.append(radii[s, :n])
        if labels is None:
            idx.append(np.full(n, s, dtype=np.int32))
        else:
            idx.append(s * C + labels[s, :n])
    return np.concatenate(cs), np.concatenate(rs), np.concatenate(idx).astype(np.int32)


def test_g1_example_literals_batched_flat_classwise():
    z = h1_cases.load("h1_g1.npz")
    H, W, C = int(z["H"]), int(z["W"]), int(z["C"])
    exp = z["expected"]
    hm = np.zeros_like(exp)
    oracle.draw_heatmap_batched(hm, z["centers"], z["radii"], z["sizes"], factor=6.0, k=1.0)
    assert np.array_equal(hm, exp)
    # known answers recorded in SURVEY.md §8c
    assert hm[0, 3, 2] == 1.0
    assert abs(float(hm[0, 0, 0]) - 0.05563799664378166) < 1e-9
    # flat
    c, r, idx = _flat_from_padded(z["centers"], z["radii"], z["sizes"])
    hm2 = np.zeros_like(exp)
    oracle.draw_heatmap_flat(hm2, c, r, idx)
    assert np.array_equal(hm2, exp)
    # class-wise
    cw = np.zeros((4, C, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(cw, z["centers"], z["radii"], z["sizes"], labels=z["labels"])
    planes = z["cw_planes"]
    mask = np.zeros((4, C), dtype=bool)
    for (s, cl), e in zip(planes.tolist(), z["cw_expected"]):
        assert np.array_equal(cw[s, cl], e)
        mask[s, cl] = True
    assert np.all(cw[~mask] == 0)


def test_g2_reference_test_recipe_padding_ignored():
    z = h1_cases.load("h1_g2.npz")
    exp = z["expected"]
    hm = np.zeros_like(exp)
    oracle.draw_heatmap_batched(hm, z["centers"], z["radii"], z["sizes"], factor=float(z["factor"]), k=float(z["k"]))
    assert np.array_equal(hm, exp)
    cw = np.zeros_like(z["cw_expected"])
    oracle.draw_heatmap_batched(cw, z["centers"], z["radii"], z["sizes"], labels=z["labels"],
                                factor=float(z["factor"]), k=float(z["k"]))
    assert np.array_equal(cw, z["cw_expected"])
    # the reference's own acceptance bar (tests/test_draw_heatmap.py:85) is MSE < 1e-3; we are exact
    c, r, idx = _flat_from_padded(z["centers"], z["radii"], z["sizes"])
    hm2 = np.zeros_like(exp)
    oracle.draw_heatmap_flat(hm2, c, r, idx, factor=float(z["factor"]), k=float(z["k"]))
    assert np.array_equal(hm2, exp)


@pytest.mark.parametrize("case", list(h1_cases.g3_cases()), ids=lambda c: c[0])
def test_g3_edge_cases(case):
    name, H, W, c, r, k, factor, base, exp = case
    hm = np.full((1, H, W), base, dtype=np.float32)
    oracle.draw_heatmap_flat(hm, c, r, np.zeros(len(r), dtype=np.int32), factor=factor, k=k)
    assert np.array_equal(hm[0], exp), name


@pytest.mark.parametrize("frame", list(h1_cases.g4_frames()), ids=lambda f: f[0])
def test_g4_full_hd_frames(frame):
    name, H, W, T, c, r, crc, sums, picks, tiles = frame
    hm = np.zeros((1, H, W), dtype=np.float32)
    oracle.draw_heatmap_batched(hm, c[None], r[None], np.array([len(r)]), clear=True)
    got_crc, got_sums = h1_cases.tile_crc_and_sums(hm[0], T)
    np.testing.assert_allclose(got_sums, sums, rtol=0, atol=1e-6)
    tx = W // T
    for p, t in zip(picks.tolist(), tiles):
        i, j = p // tx, p % tx
        assert np.array_equal(hm[0, i * T:(i + 1) * T, j * T:(j + 1) * T], t)
    assert np.array_equal(got_crc, crc), "C oracle is not bit-identical to the reference python oracle"


def test_threads_and_clear_agree():
    z = h1_cases.load("h1_g2.npz")
    a = np.full_like(z["expected"], 7.0)
    b = np.zeros_like(a)
    oracle.draw_heatmap_batched(a, z["centers"], z["radii"], z["sizes"], k=0.8, clear=True, threads=4)
    oracle.draw_heatmap_batched(b, z["centers"], z["radii"], z["sizes"], k=0.8, threads=1)
    assert np.array_equal(a, b)


def test_target_prep_front_end_cpu_matches_reference_fixture():
    """get_centers_and_radii (CPU path) against the G2 fixture: float centres/boxes -> the int32 centres/radii the
    reference helper produced (packages/draw_heatmap/tests/_test_helpers.py:20-28)."""
    import torch

    from accvlab.draw_heatmap import get_centers_and_radii

    z = h1_cases.load("h1_g2.npz")
    c, r = get_centers_and_radii(torch.from_numpy(z["centers_f"]), torch.from_numpy(z["boxes_f"]), int(z["stride"]))
    assert c.dtype == torch.int32 and r.dtype == torch.int32
    assert np.array_equal(c.numpy(), z["centers"]) and np.array_equal(r.numpy(), z["radii"])
