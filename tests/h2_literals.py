"""Hand-written known-answer vectors of the reference's batching_helpers tests, restated as DATA (inputs and
expected outputs only).  Paths are relative to packages/batching_helpers/tests/ of the reference."""
import numpy as np

BIG = 100  # the reference fills unused index slots with 100 so that reading them would fault


def gather_literal(dtype=np.float32):
    """test_batched_indexing_ops.py:31-110 — batch shape (2,3), 5 inputs x 2 channels, up to 3 indices."""
    data = np.arange(2 * 3 * 5 * 2).astype(dtype).reshape(2, 3, 5, 2)
    counts = np.array([[2, 3, 1], [3, 2, 1]], dtype=np.int64)
    idx = np.full((2, 3, 3), BIG, dtype=np.int64)
    idx[0, 0, :2] = [1, 3]
    idx[0, 1, :3] = [0, 2, 4]
    idx[0, 2, :1] = [2]
    idx[1, 0, :3] = [1, 3, 2]
    idx[1, 1, :2] = [2, 1]
    idx[1, 2, :1] = [1]
    fill = -4.1
    exp = np.full((2, 3, 3, 2), fill, dtype=dtype)
    pairs = {(0, 0): [1, 3], (0, 1): [0, 2, 4], (0, 2): [2], (1, 0): [1, 3, 2], (1, 1): [2, 1], (1, 2): [1]}
    for (a, b), js in pairs.items():
        for k, j in enumerate(js):
            exp[a, b, k] = data[a, b, j]
    grad = None
    if np.issubdtype(np.dtype(dtype), np.floating):
        grad = np.zeros_like(data)
        for (a, b), js in pairs.items():
            for j in js:
                grad[a, b, j] = np.cos(data[a, b, j])  # d/dx sum(sin(x)) at the gathered entries
    return data, idx, counts, fill, exp, grad


def mask_from_indices_literal():
    """test_batched_mask_from_indices.py:26-54."""
    idx = np.array([[2, 0, 3, BIG], [1, 2, BIG, BIG], [0, 3, 2, 4]], dtype=np.int64)
    counts = np.array([3, 2, 4], dtype=np.int64)
    exp = np.array([[1, 0, 1, 1, 0], [0, 1, 1, 0, 0], [1, 0, 1, 1, 1]], dtype=bool)
    return idx, counts, 5, exp


def compaction_literal():
    """test_batched_processing_py.py:271-323 — sizes [2,1,3,0], rows [1,3],[6],[9,10,11] / [17,19],[22],[25,26,27]."""
    d1 = np.arange(1, 17, dtype=np.int64).reshape(4, 4)
    d2 = np.arange(17, 33, dtype=np.int64).reshape(4, 4)
    mask = np.array([[1, 0, 1, 0], [0, 1, 0, 0], [1, 1, 1, 0], [0, 0, 0, 0]], dtype=bool)
    sizes = np.array([2, 1, 3, 0], dtype=np.int64)
    e1 = np.array([[1, 3, 0], [6, 0, 0], [9, 10, 11], [0, 0, 0]], dtype=np.int64)
    e2 = np.array([[17, 19, 0], [22, 0, 0], [25, 26, 27], [0, 0, 0]], dtype=np.int64)
    return mask, (d1, d2), sizes, (e1, e2)


def mapping_literal(dtype=np.float32):
    """test_batched_index_mapping_ops.py:26-110."""
    src = np.arange(1, 25).astype(dtype).reshape(3, 4, 2)
    into = -np.arange(1, 37).astype(dtype).reshape(3, 6, 2)
    src_idx = np.array([[0, 2, BIG, BIG], [1, 3, 0, BIG], [2, 1, 0, 3]], dtype=np.int64)
    dst_idx = np.array([[1, 3, BIG, BIG], [4, 2, 0, BIG], [5, 1, 0, 2]], dtype=np.int64)
    counts = np.array([2, 3, 4], dtype=np.int64)
    exp = into.copy()
    exp[0, 1], exp[0, 3] = src[0, 0], src[0, 2]
    exp[1, 4], exp[1, 2], exp[1, 0] = src[1, 1], src[1, 3], src[1, 0]
    exp[2, 5], exp[2, 1], exp[2, 0], exp[2, 2] = src[2, 2], src[2, 1], src[2, 0], src[2, 3]
    return src, src_idx, dst_idx, counts, into, exp


# ------------------------------------------------------------------------------------------------------------------
# Round 2: the remaining reference-held known answers (VERDICT r1 "What's missing" #5), restated as data.

def _cos_or_zero(a, dtype):
    """the reference writes its expected gradients as cos(x) with NaN marking "no gradient" -> 0"""
    return np.nan_to_num(np.cos(np.asarray(a, dtype=np.float64)), nan=0.0).astype(dtype)


_INV_IDX = np.array([[[0, 1, 2, BIG], [1, 2, BIG, BIG], [2, 0, 1, 4]],
                     [[3, 0, BIG, BIG], [4, 0, BIG, BIG], [1, BIG, BIG, BIG]]], dtype=np.int64)
_INV_COUNTS = np.array([[3, 2, 4], [2, 2, 1]], dtype=np.int64)


def inverse_literal(fill, dtype=np.float32):
    """test_batched_indexing_ops.py:113-202 — batch shape (2,3), 4 source slots, scatter into 5 targets of `fill`."""
    data = np.arange(1, 25).astype(dtype).reshape(2, 3, 4)
    exp = np.full((2, 3, 5), fill, dtype=dtype)
    used = np.zeros((2, 3, 4), dtype=bool)
    for a in range(2):
        for b in range(3):
            for j in range(int(_INV_COUNTS[a, b])):
                exp[a, b, _INV_IDX[a, b, j]] = data[a, b, j]
                used[a, b, j] = True
    grad = np.where(used, np.cos(data.astype(np.float64)), 0.0).astype(dtype) if np.issubdtype(np.dtype(dtype), np.floating) else None
    # spot values written out in the reference: row (0,2) -> [10, 11, 9, fill, 12]; row (1,1) -> [18, fill, fill, fill, 17]
    assert exp[0, 2].tolist()[:3] == [10, 11, 9] and exp[0, 2, 4] == 12 and exp[1, 1, 0] == 18 and exp[1, 1, 4] == 17
    return data, _INV_IDX.copy(), _INV_COUNTS.copy(), 5, exp, grad


def write_literal(dtype=np.float32):
    """test_batched_indexing_ops.py:205-339 — the same indices written into a clone of -1..-30 (2,3,5)."""
    data = np.arange(1, 25).astype(dtype).reshape(2, 3, 4)
    into = -np.arange(1, 31).astype(dtype).reshape(2, 3, 5)
    exp = into.copy()
    used = np.zeros((2, 3, 4), dtype=bool)
    kept = np.ones((2, 3, 5), dtype=bool)
    for a in range(2):
        for b in range(3):
            for j in range(int(_INV_COUNTS[a, b])):
                exp[a, b, _INV_IDX[a, b, j]] = data[a, b, j]
                used[a, b, j] = True
                kept[a, b, _INV_IDX[a, b, j]] = False
    assert exp[0, 1].tolist() == [-6, 5, 6, -9, -10] and exp[1, 2].tolist() == [-26, 21, -28, -29, -30]
    if np.issubdtype(np.dtype(dtype), np.floating):
        g_src = np.where(used, np.cos(data.astype(np.float64)), 0.0).astype(dtype)
        g_into = np.where(kept, np.cos(into.astype(np.float64)), 0.0).astype(dtype)
    else:
        g_src = g_into = None
    return data, _INV_IDX.copy(), _INV_COUNTS.copy(), into, exp, g_src, g_into


def bool_index_simple_literal(dtype=np.float32):
    """test_batched_boolean_indexing.py:33-91 — 4 samples x 5 slots x (3,2); True bits beyond a sample's size select
    padding and must be ignored."""
    data = np.arange(4 * 5 * 6).astype(dtype).reshape(4, 5, 3, 2)
    sizes = np.array([3, 5, 2, 4], dtype=np.int64)
    mask = np.zeros((4, 5), dtype=bool)
    mask[0, [0, 2, 4]] = True
    mask[1, [1, 3, 4]] = True
    mask[2, [0, 3]] = True
    mask[3, :] = True
    out_sizes = np.array([2, 3, 1, 4], dtype=np.int64)
    exp = np.zeros((4, 4, 3, 2), dtype=dtype)
    picks = {0: [0, 2], 1: [1, 3, 4], 2: [0], 3: [0, 1, 2, 3]}
    for s, js in picks.items():
        for k, j in enumerate(js):
            exp[s, k] = data[s, j]
    return data, sizes, mask, exp, out_sizes


def bool_index_multi_literal(dtype=np.float32):
    """test_batched_boolean_indexing.py:93-160 — batch shape (2,3), 4 slots x (2,3)."""
    data = np.arange(6 * 4 * 6).astype(dtype).reshape(2, 3, 4, 2, 3)
    sizes = np.array([[3, 4, 2], [1, 3, 4]], dtype=np.int64)
    mask = np.zeros((2, 3, 4), dtype=bool)
    mask[0, 0, [0, 2, 3]] = True
    mask[0, 1, [1, 2, 3]] = True
    mask[0, 2, [0, 2]] = True
    mask[1, 0, [0]] = True
    mask[1, 1, [0, 2]] = True
    mask[1, 2, [1, 3]] = True
    out_sizes = np.array([[2, 3, 1], [1, 2, 2]], dtype=np.int64)
    exp = np.zeros((2, 3, 3, 2, 3), dtype=dtype)
    picks = {(0, 0): [0, 2], (0, 1): [1, 2, 3], (0, 2): [0], (1, 0): [0], (1, 1): [0, 2], (1, 2): [1, 3]}
    for (a, b), js in picks.items():
        for k, j in enumerate(js):
            exp[a, b, k] = data[a, b, j]
    return data, sizes, mask, exp, out_sizes


def bool_write_literal(multi: bool, dtype=np.float32):
    """test_batched_boolean_indexing.py:163-242 — the inverse: the compacted rows written back through the same mask
    into a copy whose selected (valid) slots hold a marker; result == the original data."""
    data, sizes, mask, compact, out_sizes = (bool_index_multi_literal if multi else bool_index_simple_literal)(dtype)
    marker = np.nan if np.issubdtype(np.dtype(dtype), np.floating) else -999
    into = data.copy()
    valid = np.arange(mask.shape[-1]) < sizes[..., None]
    into[mask & valid] = marker
    return compact, out_sizes, mask, sizes, into, data


def pad_fill_literal(multi: bool, dtype=np.float32):
    """test_ragged_batch_set_padded_to.py:24-135 — value -10 behind each sample's size; gradient of
    sum(sin(.)) is cos(x) on valid slots, 0 on padding."""
    if multi:
        data = np.arange(1, 31).astype(dtype).reshape(2, 3, 5)
        sizes = np.array([[3, 2, 5], [4, 0, 1]], dtype=np.int64)
    else:
        data = np.arange(1, 16).astype(dtype).reshape(3, 5)
        sizes = np.array([3, 2, 5], dtype=np.int64)
    valid = np.arange(5) < sizes[..., None]
    exp = np.where(valid, data, np.asarray(-10, dtype=dtype)).astype(dtype)
    grad = np.where(valid, np.cos(data.astype(np.float64)), 0.0).astype(dtype) \
        if np.issubdtype(np.dtype(dtype), np.floating) else None
    return data, sizes, -10.0, exp, grad


def indices_from_mask_literals():
    """test_batched_processing_py.py:363-451 — (a) dense 4x6 mask, width stays 6; (b) ragged mask: True bits beyond
    the sample size are padding, width = max count."""
    m1 = np.array([[1, 0, 1, 0, 0, 1], [0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1]], dtype=bool)
    rows1 = [[0, 2, 5], [1], [], [0, 1, 2, 3, 4, 5]]
    m2 = np.array([[1, 1, 1, 0, 1], [1, 0, 1, 1, 1], [1, 1, 0, 1, 0]], dtype=bool)
    sizes2 = np.array([2, 5, 0], dtype=np.int64)
    rows2 = [[0, 1], [0, 2, 3, 4], []]
    return (m1, None, rows1, 6), (m2, sizes2, rows2, 4)


def combine_literal_shapes():
    """test_batched_processing_py.py:194-268 — shapes the reference asserts: flat list of (3,), (2,), (4,) -> (3,4);
    nested [[a,b],[c]] flattened -> (3,4); (2,4)+(3,4) -> (2,3,4); batch-shape preserving 2x3 grid of
    (n,3,4) with n = 2,1,7 / 5,3,0 -> (2,3,7,3,4)."""
    return {"flat": [3, 2, 4], "nested": [[3, 2], [4]], "extra": [(2, 4), (3, 4)],
            "grid": [[2, 1, 7], [5, 3, 0]], "grid_inner": (3, 4)}


def mapping_grads_literal(dtype=np.float32):
    """test_batched_index_mapping_ops.py:60-166 — gradients of the mapping literal: d(src) = cos(src) where a pair
    reads it, d(into) = cos(into) where nothing was written."""
    src, si, di, counts, into, exp = mapping_literal(dtype)
    g_src = np.zeros_like(src)
    g_into = np.cos(into.astype(np.float64)).astype(dtype)
    for s in range(3):
        for j in range(int(counts[s])):
            g_src[s, si[s, j]] = np.cos(src[s, si[s, j]].astype(np.float64))
            g_into[s, di[s, j]] = 0
    return g_src, g_into


def mapping_multi_batch_literal(dtype=np.float32):
    """test_batched_index_mapping_ops.py:169-341 — batch shape (2,2): src (2,2,3,2) = 1..24, into (2,2,4,2) = -1..-32."""
    src = np.arange(1, 25).astype(dtype).reshape(2, 2, 3, 2)
    into = -np.arange(1, 33).astype(dtype).reshape(2, 2, 4, 2)
    si = np.array([[[0, 2, BIG], [0, 1, 2]], [[0, BIG, BIG], [1, 2, BIG]]], dtype=np.int64)
    di = np.array([[[1, 3, BIG], [2, 0, 3]], [[2, BIG, BIG], [0, 3, BIG]]], dtype=np.int64)
    counts = np.array([[2, 3], [1, 2]], dtype=np.int64)
    exp = into.copy()
    g_src = np.zeros_like(src)
    g_into = np.cos(into.astype(np.float64)).astype(dtype)
    for a in range(2):
        for b in range(2):
            for j in range(int(counts[a, b])):
                exp[a, b, di[a, b, j]] = src[a, b, si[a, b, j]]
                g_src[a, b, si[a, b, j]] = np.cos(src[a, b, si[a, b, j]].astype(np.float64))
                g_into[a, b, di[a, b, j]] = 0
    # rows written out in the reference: (0,1) -> [[9,10],[-11,-12],[7,8],[11,12]]; (1,1) -> [[21,22],[-27,-28],[-29,-30],[23,24]]
    assert exp[0, 1].tolist() == [[9, 10], [-11, -12], [7, 8], [11, 12]]
    assert exp[1, 1].tolist() == [[21, 22], [-27, -28], [-29, -30], [23, 24]]
    return src, si, di, counts, into, exp, g_src, g_into
