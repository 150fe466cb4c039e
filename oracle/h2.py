"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path under accv-lab_amd/).

Plain numpy/python-loop restatement of the reference's ragged-batch kernels (H2).  Every function works on
numpy arrays with ONE flattened batch dimension, the indexed dimension second and arbitrary trailing data
dimensions.  Semantics follow
  packages/batching_helpers/accvlab/batching_helpers/cpp_impl/batched_indexing_access_cuda_impl.cu
    :52-113  indexing_kernel (gather / scatter-overwrite / scatter "set first, then add")
    :115-160 map_values_by_index_pairs_kernel
    :162-194 insert_const_at_indices_kernel
    :196-213 set_ragged_batch_padded_to_filler_value_kernel
  and the output-allocation rules of batched_indexing_access_cuda.cpp:54-245 (full(fill) / clone),
  batched_bool_indexing.py:195-221 (compaction), batched_processing_py.py:410-423 (combine_data).

Parity status: PINNED — tests/test_oracle_h2.py checks these functions against the hand-written expected
tensors of the reference's own tests (restated as data in tests/h2_literals.py, with file:line).
"""
from __future__ import annotations

import numpy as np


def _wrap(j: int, width: int) -> int:
    """Negative indices wrap once (cu:75-77); anything still outside is an error (device assert, cu:79)."""
    if j < 0:
        j += width
    if j < 0 or j >= width:
        raise IndexError(f"index {j} out of range for width {width}")
    return j


def gather(data: np.ndarray, idx: np.ndarray, counts: np.ndarray, fill) -> np.ndarray:
    """out[i, j] = data[i, idx[i, j]] for j < counts[i]; `fill` elsewhere.  (forward, cpp:54-86)"""
    b, w_idx = idx.shape
    out = np.full((b, w_idx) + data.shape[2:], fill, dtype=data.dtype)
    for i in range(b):
        for j in range(min(int(counts[i]), w_idx)):
            out[i, j] = data[i, _wrap(int(idx[i, j]), data.shape[1])]
    return out


def scatter_new(to_insert: np.ndarray, idx: np.ndarray, counts: np.ndarray, num_targets: int, fill,
                accumulate: bool) -> np.ndarray:
    """Fresh full(fill) tensor; out[i, idx[i, j]] = / += to_insert[i, j].  With `accumulate` the first write to a
    slot replaces the filler and later ones add (cu:39-50).  (backward_new_tensor, cpp:88-120)"""
    b, w_idx = idx.shape
    out = np.full((b, num_targets) + to_insert.shape[2:], fill, dtype=to_insert.dtype)
    touched = np.zeros((b, num_targets), dtype=bool)
    for i in range(b):
        for j in range(min(int(counts[i]), w_idx)):
            o = _wrap(int(idx[i, j]), num_targets)
            if accumulate and touched[i, o]:
                out[i, o] = out[i, o] + to_insert[i, j]
            else:
                out[i, o] = to_insert[i, j]
            touched[i, o] = True
    return out


def scatter_insert(to_insert: np.ndarray, idx: np.ndarray, counts: np.ndarray, into: np.ndarray) -> np.ndarray:
    """Clone of `into` with out[i, idx[i, j]] = to_insert[i, j].  (backward_insert, cpp:122-146)"""
    out = into.copy()
    for i in range(idx.shape[0]):
        for j in range(min(int(counts[i]), idx.shape[1])):
            out[i, _wrap(int(idx[i, j]), into.shape[1])] = to_insert[i, j]
    return out


def insert_const(value, idx: np.ndarray, counts: np.ndarray, into: np.ndarray) -> np.ndarray:
    """Clone of `into` with the constant at the indexed slots.  (backward_insert_const, cpp:148-168)"""
    out = into.copy()
    for i in range(idx.shape[0]):
        for j in range(min(int(counts[i]), idx.shape[1])):
            out[i, _wrap(int(idx[i, j]), into.shape[1])] = value
    return out


def map_pairs(src: np.ndarray, src_idx: np.ndarray, dst_idx: np.ndarray, counts: np.ndarray, into: np.ndarray,
              accumulate: bool = False) -> np.ndarray:
    """Clone of `into` with out[i, dst_idx[i, j]] = / += src[i, src_idx[i, j]] (first write replaces).
    (map_values_by_index_pairs, cpp:170-200)"""
    out = into.copy()
    touched = np.zeros(into.shape[:2], dtype=bool)
    for i in range(src_idx.shape[0]):
        for j in range(min(int(counts[i]), src_idx.shape[1])):
            s = _wrap(int(src_idx[i, j]), src.shape[1])
            o = _wrap(int(dst_idx[i, j]), into.shape[1])
            if accumulate and touched[i, o]:
                out[i, o] = out[i, o] + src[i, s]
            else:
                out[i, o] = src[i, s]
            touched[i, o] = True
    return out


def mask_from_indices(idx: np.ndarray, counts: np.ndarray, num_targets: int) -> np.ndarray:
    """(get_mask_from_indices, cpp:202-228)"""
    return insert_const(True, idx, counts, np.zeros((idx.shape[0], num_targets), dtype=bool))


def pad_fill(data: np.ndarray, counts: np.ndarray, value) -> np.ndarray:
    """Copy with data[i, j] = value for j >= counts[i].  (set_ragged_batch_padded_to_filler_value, cu:196-213;
    CPU twin batched_indexing_access_cpu_impl.cpp:27-44)"""
    out = data.copy()
    for i in range(data.shape[0]):
        out[i, max(0, int(counts[i])):] = value
    return out


def bool_compact(data: np.ndarray, mask: np.ndarray, valid=None):
    """Per row, the entries whose mask is True, in order, left-aligned, zeros behind; width = max count.
    Returns (padded, sizes int64).  (batched_bool_indexing.py:195-221; get_compact_lists, batched_processing_py.py
    :243-268)"""
    b, m = mask.shape
    rows = []
    for i in range(b):
        lim = m if valid is None else max(0, min(m, int(valid[i])))
        rows.append([j for j in range(lim) if mask[i, j]])
    sizes = np.array([len(r) for r in rows], dtype=np.int64)
    width = int(sizes.max()) if b else 0
    out = np.zeros((b, width) + data.shape[2:], dtype=data.dtype)
    for i, r in enumerate(rows):
        for k, j in enumerate(r):
            out[i, k] = data[i, j]
    return out, sizes


def indices_from_mask(mask: np.ndarray, valid=None):
    """(get_indices_from_mask, batched_processing_py.py:577-628)"""
    cols = np.broadcast_to(np.arange(mask.shape[1], dtype=np.int64), mask.shape)
    return bool_compact(cols, mask, valid)


def bool_write(to_write: np.ndarray, write_sizes: np.ndarray, mask: np.ndarray, into: np.ndarray, valid=None) -> np.ndarray:
    """Clone of `into`; the k-th True of mask[i] receives to_write[i, k].  (batched_bool_indexing_write,
    batched_bool_indexing.py:350-359)"""
    out = into.copy()
    for i in range(mask.shape[0]):
        lim = mask.shape[1] if valid is None else max(0, min(mask.shape[1], int(valid[i])))
        k = 0
        for j in range(lim):
            if mask[i, j] and k < int(write_sizes[i]):
                out[i, j] = to_write[i, k]
                k += 1
    return out


def combine(samples):
    """list of (n_i, *d) arrays -> (padded [B, max n, *d] zero-padded, sizes).  (combine_data, :410-423)"""
    b = len(samples)
    width = max([s.shape[0] for s in samples] + [0])
    proto = next((s for s in samples if s.size > 0), samples[0])
    out = np.zeros((b, width) + proto.shape[1:], dtype=proto.dtype)
    sizes = np.zeros(b, dtype=np.int64)
    for i, s in enumerate(samples):
        n = min(s.shape[0], s.size)
        sizes[i] = n
        if n:
            out[i, :n] = s
    return out, sizes
