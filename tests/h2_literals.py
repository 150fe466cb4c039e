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
