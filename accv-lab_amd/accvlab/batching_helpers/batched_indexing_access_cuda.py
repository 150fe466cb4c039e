"""Drop-in for the reference's extension module ``accvlab.batching_helpers.batched_indexing_access_cuda``
(pybind table: cpp_impl/batched_indexing_access_cuda.cpp:247-265): same seven functions, same argument
names/defaults, same return conventions (new tensors except the ``_in_place`` op), same validation
(cpp:54-245 with the CHECK_* macros of batched_indexing_access_helpers.h:60-148 -> RuntimeError).

Every function validates on the host, allocates its result with torch (``full`` / ``clone`` exactly
where the reference does) and enqueues ONE OR TWO kernels of libaccv_hip.so on torch's current stream.
There is no CPU path in this module (the reference's is CUDA-only too).
"""
from __future__ import annotations

import functools
import struct
from typing import Optional

import torch

from .. import _amd_native as _nat

try:  # C++ fast path of the hottest operators (csrc_host/bh_host.cpp); the python code below is complete without it
    from . import _bh_host as _bh
except ImportError:  # pragma: no cover
    _bh = None
_native_bound = False


def _native():
    """_bh_host with the C-ABI addresses bound (same library handle as the ctypes binding), or None."""
    global _native_bound
    if _bh is None or not hasattr(_bh, "bind_native") or _nat.NO_HOST_FASTPATH:
        return None
    if not _native_bound:
        import ctypes

        h = _nat.ctypes_lib()
        addr = lambda name: ctypes.cast(getattr(h, name), ctypes.c_void_p).value  # noqa: E731
        _bh.bind_native(addr("accv_ragged_gather"), addr("accv_ragged_scatter"), addr("accv_ragged_gather_fill"),
                        addr("accv_last_error"))
        if hasattr(_bh, "bind_mask_to_indices"):
            _bh.bind_mask_to_indices(addr("accv_ragged_mask_to_indices_ws"), addr("accv_ragged_mask_to_indices_workspace_bytes"))
        _native_bound = True
    return _bh


_COPY_DTYPES = {torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64}
_ACC_CODE = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3, torch.float16: 4, torch.bfloat16: 5}
_INDEX_DTYPES = (torch.int32, torch.int64)


def _contig(t: torch.Tensor, name: str) -> None:
    if not (t.is_contiguous()):
        raise RuntimeError(f"{name} must be contiguous")


def _same_cuda_device(*named) -> None:
    name0, first = named[0]
    if not (first.is_cuda):
        raise RuntimeError(f"{name0} must be a CUDA tensor")
    for _, t in named[1:]:
        if not (t.device == first.device):
            raise RuntimeError("All input tensors must be on the same device")


def _dims_at_least(t, n, name):
    if t.numel() != 0:
        if not (t.dim() >= n):
            raise RuntimeError(f"{name} must have at least {n} dimensions")


def _dims_exact(t, n, name):
    if t.numel() != 0:
        if not (t.dim() == n):
            raise RuntimeError(f"{name} must have {n} dimensions")


def _match_first(a, b, n, na, nb_):
    if a.numel() == 0 and b.numel() == 0:
        return
    if not (a.dim() >= n and b.dim() >= n):
        raise RuntimeError(f"{na} and {nb_} must have at least {n} dimensions")
    for i in range(n):
        if not (a.size(i) == b.size(i)):
            raise RuntimeError(f"{na} and {nb_} must have the same size in dimension {i}")


def _match_except(a, b, dim, na, nb_):
    if a.numel() == 0 and b.numel() == 0:
        return
    if not (a.dim() == b.dim()):
        raise RuntimeError(f"{na} and {nb_} must have the same number of dimensions")
    for i in range(a.dim()):
        if i != dim:
            if not (a.size(i) == b.size(i)):
                raise RuntimeError(f"{na} and {nb_} must have the same size in dimension {i}")


def _match_all(a, b, na, nb_):
    if a.numel() == 0 and b.numel() == 0:
        return
    if not (a.dim() == b.dim()):
        raise RuntimeError(f"{na} and {nb_} must have the same number of dimensions")
    if not (tuple(a.shape) == tuple(b.shape)):
        raise RuntimeError(f"{na} and {nb_} must have the same size")


def _index_dtype(t, name) -> int:
    if not (t.dtype in _INDEX_DTYPES):
        raise RuntimeError(f"{name}: index tensors must be int32 or int64, got {t.dtype}")
    return 1 if t.dtype == torch.int64 else 0


def _data_dtype(t, name, allow_bool=False) -> None:
    ok = t.dtype in _COPY_DTYPES or (allow_bool and t.dtype == torch.bool)
    if not (ok):
        raise RuntimeError(f"{name}: unsupported data type {t.dtype}")


@functools.lru_cache(maxsize=256)
def element_bits(value, dtype: torch.dtype) -> int:
    """Byte pattern (as an unsigned integer, little endian) of ``value`` converted to ``dtype`` with the
    conversion rules of ``static_cast<scalar_t>(double)`` used by the reference."""
    # normalise to a plain python number first: the cache below must never key on a mutable object (0-dim tensors, numpy
    # scalars) whose value can change under the same identity
    if not isinstance(value, (bool, int, float)):
        value = float(value)
    return _element_bits(value, dtype)


@functools.lru_cache(maxsize=256)
def _element_bits(value, dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return struct.unpack("<I", struct.pack("<f", float(value)))[0]
    if dtype == torch.float64:
        return struct.unpack("<Q", struct.pack("<d", float(value)))[0]
    if dtype == torch.float16:
        return struct.unpack("<H", struct.pack("<e", float(value)))[0]
    if dtype == torch.bfloat16:
        return int(torch.tensor(float(value), dtype=torch.bfloat16).view(torch.int16).item()) & 0xFFFF
    if dtype == torch.bool:
        return 1 if value else 0
    bits = {torch.int8: 8, torch.uint8: 8, torch.int16: 16, torch.int32: 32, torch.int64: 64}.get(dtype)
    if bits is None:
        raise RuntimeError(f"unsupported data type {dtype}")
    return int(value) & ((1 << bits) - 1)


def _row_elems(t: torch.Tensor, first_data_dim: int) -> int:
    n = 1
    for s in t.shape[first_data_dim:]:
        n *= int(s)
    return n


def _batch_numel(counts: torch.Tensor) -> int:
    return int(counts.numel())


def _call(status: int, what: str) -> None:
    _nat.check(status, what)


def _stream(t: torch.Tensor) -> int:
    return _nat.stream_ptr(t.device)


# ------------------------------------------------------------------------------------------------ forward
def forward(input_data: torch.Tensor, input_indices: torch.Tensor, input_nums_indices: torch.Tensor,
            fill_value: float = 0.0) -> torch.Tensor:
    """``res[i, j] = input_data[i, input_indices[i, j]]`` for ``j < input_nums_indices[i]``, ``fill_value``
    elsewhere (cpp:54-86, kernel cu:52-113 forward direction)."""
    nat = _native()
    if nat is not None and type(input_data) is torch.Tensor and input_data.dtype in _COPY_DTYPES:
        # checks, allocation, stream and the C-ABI call in C++; it declines (None) whatever the python path below must
        # diagnose or handle (wrong devices / dtypes / shapes, empty results)
        res = nat.forward_gather_fill(input_data, input_indices, input_nums_indices, element_bits(fill_value, input_data.dtype))
        if res is not None:
            return res.requires_grad_(True) if input_data.requires_grad else res
    _contig(input_data, "input_data")
    _contig(input_indices, "input_indices")
    _contig(input_nums_indices, "input_nums_indices")
    _same_cuda_device(("input_data", input_data), ("input_indices", input_indices),
                      ("input_nums_indices", input_nums_indices))
    _dims_at_least(input_nums_indices, 1, "input_nums_indices")
    nb = input_nums_indices.dim()
    _dims_at_least(input_indices, 1, "input_indices")
    _dims_at_least(input_data, nb + 1, "input_data")
    _match_first(input_data, input_indices, nb, "input_data", "input_indices")
    _match_first(input_indices, input_nums_indices, nb, "input_indices", "input_nums_indices")
    if not (input_indices.dim() >= nb + 1):
        raise RuntimeError(f"input_indices must have at least {nb + 1} dimensions")
    _data_dtype(input_data, "input_data")
    res_size = list(input_indices.shape[:nb + 1]) + list(input_data.shape[nb + 1:])
    if input_indices.numel() == 0 or 0 in res_size:
        return torch.full(res_size, fill_value, dtype=input_data.dtype, device=input_data.device,
                          requires_grad=input_data.requires_grad)
    # the kernel writes every element of the result (gathered rows and the filler): no torch.full pass
    res = torch.empty(res_size, dtype=input_data.dtype, device=input_data.device,
                      requires_grad=input_data.requires_grad)
    batch = _batch_numel(input_nums_indices)
    w_idx = input_indices.size(nb)
    esz = input_data.element_size()
    row_bytes = _row_elems(input_data, nb + 1) * esz
    with _nat.device_guard(input_data.device):
        _call(_nat.lib().accv_ragged_gather_fill(
            input_data.data_ptr(), res.data_ptr(), input_indices.data_ptr(), input_nums_indices.data_ptr(), batch,
            input_data.size(nb), w_idx, w_idx, row_bytes, element_bits(fill_value, input_data.dtype), esz,
            _index_dtype(input_indices, "input_indices"), _index_dtype(input_nums_indices, "input_nums_indices"), None,
            _stream(input_data)), "forward")
    return res


def _scatter_into(res, to_insert, indices, counts, nb, accumulate, clear_first, what):
    batch = _batch_numel(counts)
    w_idx = indices.size(nb)
    elems = _row_elems(to_insert, nb + 1)
    esz = to_insert.element_size()
    ii, ci = _index_dtype(indices, "input_indices"), _index_dtype(counts, "input_nums_indices")
    lib = _nat.lib()
    with _nat.device_guard(res.device):
        s = _stream(res)
        if not accumulate:
            _call(lib.accv_ragged_scatter(to_insert.data_ptr(), res.data_ptr(), indices.data_ptr(), counts.data_ptr(),
                                          batch, w_idx, w_idx, res.size(nb), elems * esz, ii, ci, None, s), what)
        else:
            if not (to_insert.dtype in _ACC_CODE):
                raise RuntimeError(f"{what}: accumulation is not supported for {to_insert.dtype}")
            if clear_first:
                _call(lib.accv_ragged_insert_const(res.data_ptr(), indices.data_ptr(), counts.data_ptr(), batch, w_idx,
                                                   w_idx, res.size(nb), elems * esz, 0, esz, ii, ci, None, s), what)
            _call(lib.accv_ragged_accumulate(to_insert.data_ptr(), res.data_ptr(), None, indices.data_ptr(),
                                             counts.data_ptr(), batch, w_idx, w_idx, w_idx, res.size(nb), elems,
                                             _ACC_CODE[to_insert.dtype], ii, ci, None, s), what)


def backward_new_tensor(to_insert: torch.Tensor, input_indices: torch.Tensor, input_nums_indices: torch.Tensor,
                        input_num_targets: int, fill_value: float = 0.0,
                        backward_accumulate: bool = True) -> torch.Tensor:
    """Fresh ``full(fill_value)`` tensor of width ``input_num_targets`` with
    ``res[i, input_indices[i, j]] (+)= to_insert[i, j]`` (cpp:88-120).  With ``backward_accumulate`` several
    slots pointing at one target are summed (the first write replaces the filler: cu:39-50)."""
    _contig(to_insert, "to_insert")
    _contig(input_indices, "input_indices")
    _contig(input_nums_indices, "input_nums_indices")
    _same_cuda_device(("to_insert", to_insert), ("input_indices", input_indices),
                      ("input_nums_indices", input_nums_indices))
    _dims_at_least(input_nums_indices, 1, "input_nums_indices")
    nb = input_nums_indices.dim()
    _dims_at_least(to_insert, nb + 1, "to_insert")
    _dims_exact(input_indices, nb + 1, "input_indices")
    _match_first(to_insert, input_indices, nb, "to_insert", "input_indices")
    _match_first(input_indices, input_nums_indices, nb, "input_indices", "input_nums_indices")
    if not (to_insert.dim() >= nb + 1):
        raise RuntimeError(f"to_insert must have at least {nb + 1} dimensions")
    _data_dtype(to_insert, "to_insert")
    shape = list(to_insert.shape)
    shape[nb] = int(input_num_targets)
    res = torch.full(shape, fill_value, dtype=to_insert.dtype, device=to_insert.device,
                     requires_grad=to_insert.requires_grad)
    if input_indices.numel() == 0 or to_insert.numel() == 0 or res.numel() == 0:
        return res
    _scatter_into(res, to_insert, input_indices, input_nums_indices, nb, backward_accumulate,
                  clear_first=(fill_value != 0.0), what="backward_new_tensor")
    return res


def backward_insert(to_insert: torch.Tensor, input_indices: torch.Tensor, input_nums_indices: torch.Tensor,
                    to_insert_into: torch.Tensor) -> torch.Tensor:
    """Clone of ``to_insert_into`` with ``res[i, input_indices[i, j]] = to_insert[i, j]`` (cpp:122-146)."""
    for n, t in (("to_insert", to_insert), ("input_indices", input_indices),
                 ("input_nums_indices", input_nums_indices), ("to_insert_into", to_insert_into)):
        _contig(t, n)
    _same_cuda_device(("to_insert", to_insert), ("input_indices", input_indices),
                      ("input_nums_indices", input_nums_indices), ("to_insert_into", to_insert_into))
    if not (to_insert.dtype == to_insert_into.dtype):
        raise RuntimeError("Same dtype required for `to_insert` and `to_insert_into`")
    _dims_at_least(input_nums_indices, 1, "input_nums_indices")
    nb = input_nums_indices.dim()
    _dims_at_least(to_insert, nb + 1, "to_insert")
    _dims_exact(input_indices, nb + 1, "input_indices")
    _match_first(to_insert, input_indices, nb, "to_insert", "input_indices")
    _match_first(input_indices, input_nums_indices, nb, "input_indices", "input_nums_indices")
    _match_except(to_insert, to_insert_into, nb, "to_insert", "to_insert_into")
    _data_dtype(to_insert, "to_insert")
    res = to_insert_into.clone()
    if input_indices.numel() == 0 or to_insert.numel() == 0 or res.numel() == 0:
        return res
    _scatter_into(res, to_insert, input_indices, input_nums_indices, nb, False, False, "backward_insert")
    return res


def backward_insert_const(to_insert: float, input_indices: torch.Tensor, input_nums_indices: torch.Tensor,
                          to_insert_into: torch.Tensor) -> torch.Tensor:
    """Clone of ``to_insert_into`` with the constant written at the indexed slots (cpp:148-168)."""
    for n, t in (("input_indices", input_indices), ("input_nums_indices", input_nums_indices),
                 ("to_insert_into", to_insert_into)):
        _contig(t, n)
    _same_cuda_device(("input_indices", input_indices), ("input_nums_indices", input_nums_indices),
                      ("to_insert_into", to_insert_into))
    _dims_at_least(input_nums_indices, 1, "input_nums_indices")
    nb = input_nums_indices.dim()
    _dims_at_least(to_insert_into, nb + 1, "to_insert_into")
    _dims_exact(input_indices, nb + 1, "input_indices")
    _match_first(input_indices, input_nums_indices, nb, "input_indices", "input_nums_indices")
    _data_dtype(to_insert_into, "to_insert_into")
    res = to_insert_into.clone()
    if input_indices.numel() == 0 or res.numel() == 0:
        return res
    _insert_const(res, to_insert, input_indices, input_nums_indices, nb, "backward_insert_const")
    return res


def _insert_const(res, value, indices, counts, nb, what):
    esz = res.element_size()
    w_idx = indices.size(nb)
    with _nat.device_guard(res.device):
        _call(_nat.lib().accv_ragged_insert_const(
            res.data_ptr(), indices.data_ptr(), counts.data_ptr(), _batch_numel(counts), w_idx, w_idx, res.size(nb),
            _row_elems(res, nb + 1) * esz, element_bits(value, res.dtype), esz, _index_dtype(indices, "input_indices"),
            _index_dtype(counts, "input_nums_indices"), None, _stream(res)), what)


def map_values_by_index_pairs(input_data: torch.Tensor, input_indices: torch.Tensor, output_indices: torch.Tensor,
                              nums_indices: torch.Tensor, to_insert_into: torch.Tensor,
                              backward_accumulate: bool = False) -> torch.Tensor:
    """Clone of ``to_insert_into`` with ``res[i, output_indices[i, j]] (+)= input_data[i, input_indices[i, j]]``
    (cpp:170-200, kernel cu:115-160).  Index and count tensors share one integer dtype (cu:462-477)."""
    for n, t in (("input_data", input_data), ("input_indices", input_indices), ("output_indices", output_indices),
                 ("nums_indices", nums_indices), ("to_insert_into", to_insert_into)):
        _contig(t, n)
    _same_cuda_device(("input_data", input_data), ("input_indices", input_indices),
                      ("output_indices", output_indices), ("nums_indices", nums_indices),
                      ("to_insert_into", to_insert_into))
    if not (input_data.dtype == to_insert_into.dtype):
        raise RuntimeError("Same dtype required for `input_data` and `to_insert_into`")
    _dims_at_least(nums_indices, 1, "nums_indices")
    nb = nums_indices.dim()
    _dims_at_least(input_data, nb + 1, "input_data")
    _dims_exact(input_indices, nb + 1, "input_indices")
    _dims_exact(output_indices, nb + 1, "output_indices")
    _match_first(input_data, input_indices, nb, "input_data", "input_indices")
    _match_all(input_indices, output_indices, "input_indices", "output_indices")
    _match_first(input_indices, nums_indices, nb, "input_indices", "nums_indices")
    _match_except(input_data, to_insert_into, nb, "input_data", "to_insert_into")
    _data_dtype(to_insert_into, "to_insert_into")
    if not (input_indices.dtype == output_indices.dtype):
        raise RuntimeError("input_indices and output_indices must have the same dtype")
    res = to_insert_into.clone()
    if input_indices.numel() == 0 or res.numel() == 0 or input_data.numel() == 0:
        return res
    ii = _index_dtype(input_indices, "input_indices")
    ci = _index_dtype(nums_indices, "nums_indices")
    batch, w_idx = _batch_numel(nums_indices), input_indices.size(nb)
    elems, esz = _row_elems(res, nb + 1), res.element_size()
    lib = _nat.lib()
    with _nat.device_guard(res.device):
        s = _stream(res)
        if not backward_accumulate:
            _call(lib.accv_ragged_map_pairs(input_data.data_ptr(), res.data_ptr(), input_indices.data_ptr(),
                                            output_indices.data_ptr(), nums_indices.data_ptr(), batch,
                                            input_data.size(nb), w_idx, w_idx, res.size(nb), elems * esz, ii, ci, None,
                                            s), "map_values_by_index_pairs")
        else:
            if not (res.dtype in _ACC_CODE):
                raise RuntimeError(f"map_values_by_index_pairs: accumulation is not supported for {res.dtype}")
            _call(lib.accv_ragged_insert_const(res.data_ptr(), output_indices.data_ptr(), nums_indices.data_ptr(), batch,
                                               w_idx, w_idx, res.size(nb), elems * esz, 0, esz, ii, ci, None, s),
                  "map_values_by_index_pairs")
            _call(lib.accv_ragged_accumulate(input_data.data_ptr(), res.data_ptr(), input_indices.data_ptr(),
                                             output_indices.data_ptr(), nums_indices.data_ptr(), batch,
                                             input_data.size(nb), w_idx, w_idx, res.size(nb), elems,
                                             _ACC_CODE[res.dtype], ii, ci, None, s), "map_values_by_index_pairs")
    return res


def get_mask_from_indices(indices: torch.Tensor, nums_indices: torch.Tensor, num_targets: int) -> torch.Tensor:
    """bool ``[*batch, num_targets]`` with True at ``indices[i, :nums_indices[i]]`` (cpp:202-228)."""
    _contig(indices, "indices")
    _contig(nums_indices, "nums_indices")
    _same_cuda_device(("indices", indices), ("nums_indices", nums_indices))
    _dims_at_least(nums_indices, 1, "nums_indices")
    nb = nums_indices.dim()
    _dims_exact(indices, nb + 1, "indices")
    _match_first(indices, nums_indices, nb, "indices", "nums_indices")
    res = torch.zeros(list(nums_indices.shape) + [int(num_targets)], dtype=torch.bool, device=indices.device)
    if indices.numel() == 0 or res.numel() == 0:
        return res
    _insert_const(res, True, indices, nums_indices, nb, "get_mask_from_indices")
    return res


def set_ragged_batch_padded_to_filler_value_in_place(data: torch.Tensor, nums_valid_entries: torch.Tensor,
                                                     filler_value: float) -> None:
    """``data[i, j, ...] = filler_value`` for ``j >= nums_valid_entries[i]``, in place (cpp:230-245)."""
    _contig(data, "data")
    _contig(nums_valid_entries, "nums_valid_entries")
    _same_cuda_device(("data", data), ("nums_valid_entries", nums_valid_entries))
    _dims_at_least(nums_valid_entries, 1, "nums_valid_entries")
    nb = nums_valid_entries.dim()
    _dims_at_least(data, nb + 1, "data")
    _match_first(data, nums_valid_entries, nb, "data", "nums_valid_entries")
    _data_dtype(data, "data", allow_bool=True)
    if data.numel() == 0:
        return
    esz = data.element_size()
    with _nat.device_guard(data.device):
        _call(_nat.lib().accv_ragged_pad_fill(
            data.data_ptr(), nums_valid_entries.data_ptr(), _batch_numel(nums_valid_entries), data.size(nb),
            _row_elems(data, nb + 1) * esz, element_bits(filler_value, data.dtype), esz,
            _index_dtype(nums_valid_entries, "nums_valid_entries"), _stream(data)),
            "set_ragged_batch_padded_to_filler_value_in_place")


# ------------------------------------------------------------------------------------------------ extensions
def mask_to_indices(mask: torch.Tensor, valid_counts: Optional[torch.Tensor] = None):
    """(extension, no reference counterpart in the native layer) positions of the True entries of every row
    of a 2-D mask, in order, as int64 ``[B, M]`` zero-filled behind, plus int64 counts ``[B]`` — the
    wave-ballot compaction that replaces torch boolean indexing in batched_bool_indexing."""
    nat = _native()
    if nat is not None and hasattr(nat, "mask_to_indices") and mask.dtype == torch.bool:
        res = nat.mask_to_indices(mask, valid_counts)      # checks + allocations + launch in C++; None = declined
        if res is not None:
            return res
    if not (mask.is_cuda):
        raise RuntimeError("mask must be a CUDA tensor")
    if not (mask.dim() == 2):
        raise RuntimeError("mask must be 2-D")
    m = mask if mask.dtype == torch.bool else mask != 0
    m = m.contiguous()
    b, w = m.shape
    idx = torch.empty((b, w), dtype=torch.int64, device=m.device)
    sizes = torch.empty((b,), dtype=torch.int64, device=m.device)
    vc_ptr, vc64 = None, 0
    if valid_counts is not None:
        if not (valid_counts.device == m.device and valid_counts.numel() == b):
            raise RuntimeError("valid_counts must match the mask rows")
        valid_counts = valid_counts.contiguous()
        vc_ptr, vc64 = valid_counts.data_ptr(), _index_dtype(valid_counts, "valid_counts")
    if b > 0:
        lib = _nat.lib()
        with _nat.device_guard(m.device):
            # few, very wide rows take the segmented two-pass kernels, which need a few KB of workspace
            ws_bytes = lib.accv_ragged_mask_to_indices_workspace_bytes(b, w) if w >= 8192 else 0   # (0 below 2 segments)
            if ws_bytes:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=m.device)
                _call(lib.accv_ragged_mask_to_indices_ws(m.data_ptr(), vc_ptr, vc64, b, w, idx.data_ptr(), sizes.data_ptr(),
                                                         ws.data_ptr(), ws_bytes, _stream(m)), "mask_to_indices")
            else:
                _call(lib.accv_ragged_mask_to_indices(m.data_ptr(), vc_ptr, vc64, b, w, idx.data_ptr(),
                                                      sizes.data_ptr(), _stream(m)), "mask_to_indices")
    return idx, sizes


def gather_rows(src: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor, w_idx: int, out: torch.Tensor) -> None:
    """(extension) ``out[i, j] = src[i, indices[i, j]]`` for j < counts[i], j < w_idx, where ``indices`` may be
    wider than ``w_idx`` (row stride = indices.size(1)); any dtype incl. bool; single batch dimension."""
    nat = _native()
    if nat is not None and nat.gather_rows(src, indices, counts, int(w_idx), out):
        return
    if not (src.is_contiguous() and out.is_contiguous() and indices.is_contiguous()):
        raise RuntimeError("gather_rows: contiguous tensors required")
    if out.numel() == 0 or w_idx == 0:
        return
    row_bytes = _row_elems(src, 2) * src.element_size()
    with _nat.device_guard(src.device):
        _call(_nat.lib().accv_ragged_gather(src.data_ptr(), out.data_ptr(), indices.data_ptr(), counts.data_ptr(),
                                            src.size(0), src.size(1), int(w_idx), indices.size(1), row_bytes,
                                            _index_dtype(indices, "indices"), _index_dtype(counts, "counts"), None,
                                            _stream(src)), "gather_rows")


def scatter_rows(src: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor, w_idx: int, out: torch.Tensor) -> None:
    """(extension) ``out[i, indices[i, j]] = src[i, j]`` for j < counts[i], j < w_idx (src width == w_idx)."""
    nat = _native()
    if nat is not None and nat.scatter_rows(src, indices, counts, int(w_idx), out):
        return
    if not (src.is_contiguous() and out.is_contiguous() and indices.is_contiguous()):
        raise RuntimeError("scatter_rows: contiguous tensors required")
    if src.numel() == 0 or w_idx == 0:
        return
    row_bytes = _row_elems(src, 2) * src.element_size()
    with _nat.device_guard(src.device):
        _call(_nat.lib().accv_ragged_scatter(src.data_ptr(), out.data_ptr(), indices.data_ptr(), counts.data_ptr(),
                                             src.size(0), int(w_idx), indices.size(1), out.size(1), row_bytes,
                                             _index_dtype(indices, "indices"), _index_dtype(counts, "counts"), None,
                                             _stream(src)), "scatter_rows")


def pack_rows(flat: torch.Tensor, offsets: torch.Tensor, sizes: torch.Tensor, width: int) -> torch.Tensor:
    """(extension) padded ``[B, width, *inner]`` from ``flat [total, *inner]`` — combine_data on the device."""
    if not (flat.is_cuda and flat.is_contiguous()):
        raise RuntimeError("pack_rows: flat must be a contiguous CUDA tensor")
    b = sizes.numel()
    out = torch.empty((b, int(width)) + tuple(flat.shape[1:]), dtype=flat.dtype, device=flat.device)
    if out.numel() == 0:
        return out
    if flat.numel() == 0:
        return out.zero_()
    row_bytes = _row_elems(flat, 1) * flat.element_size()
    with _nat.device_guard(flat.device):
        _call(_nat.lib().accv_ragged_pack(flat.data_ptr(), out.data_ptr(), offsets.data_ptr(), sizes.data_ptr(), b,
                                          int(width), row_bytes, 0, _stream(flat)), "pack_rows")
    return out
