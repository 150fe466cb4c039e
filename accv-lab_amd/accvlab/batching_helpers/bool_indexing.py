"""Ragged compaction by boolean mask and its inverse.

Same public contract as the reference (batched_bool_indexing.py:90-234, 237-368;
batched_processing_py.py:200-272 get_compact_lists, :577-628 get_indices_from_mask).  On CUDA tensors the
boolean indexing of the reference (nonzero + gather + scatter, two hidden syncs) is replaced by ONE
wave-ballot compaction kernel (mask -> ordered indices + counts) followed by the ragged gather/scatter
kernel; only the read-back of the maximum count, which decides the output SHAPE, synchronises.  CPU tensors
take the torch path (the reference runs there too).
"""
from __future__ import annotations

from typing import Any, List, NamedTuple, Optional, Sequence, Union

import torch

from . import batched_indexing_access_cuda as _ext
from .indexing import BatchedIndexingAccess, BatchedInverseIndexingAccessInsert
from .ragged import RaggedBatch

_AUTOGRAD_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int32, torch.int64)


def _check_pair(data, mask) -> None:
    """Shape compatibility of the thing indexed and its mask (AssertionError on mismatch)."""
    d_rb, m_rb = isinstance(data, RaggedBatch), isinstance(mask, RaggedBatch)
    if d_rb and m_rb:
        assert data.num_batch_dims == mask.num_batch_dims, "Data and mask must have the same number of batch dimensions"
        assert data.batch_shape == mask.batch_shape, "Data and mask must have the same batch shape"
        assert data.max_sample_size == mask.max_sample_size, "Data and mask must have the same maximum sample size"
    elif d_rb:
        assert data.num_batch_dims == 1, "Data must have exactly one batch dimension if mask is a tensor"
        assert data.batch_shape[0] == mask.shape[0], "Data and mask must have the same number of samples"
        assert data.max_sample_size == mask.shape[1], \
            "Maximum sample size of data must correspond to `input_mask.shape[1]` if the mask is a tensor"
    elif m_rb:
        assert mask.num_batch_dims == 1, "Mask must have exactly one batch dimension if input data is a tensor"
        assert mask.batch_shape[0] == data.shape[0], "Mask and data must have the same number of samples"
        assert mask.max_sample_size == data.shape[1], \
            "Maximum sample size of mask must correspond to `input_data.shape[1]` if the input data is a tensor"
    else:
        assert data.shape[0] == mask.shape[0], "Data and mask must have the same number of samples"
        assert data.shape[1] == mask.shape[1], "Data and mask must have the same maximum sample size"


def _flat_mask_and_valid(mask, data):
    """2-D bool mask tensor plus the per-row count of columns that may be looked at (None = all)."""
    valid = None
    if isinstance(mask, RaggedBatch):
        valid = mask.sample_sizes
        if mask.num_batch_dims > 1:
            mask = mask.flatten_batch_dims()
            valid = mask.sample_sizes
        m = mask.tensor
    else:
        m = mask
        if isinstance(data, RaggedBatch):
            valid = data.sample_sizes.reshape(-1)
    return m, (valid.reshape(-1) if valid is not None else None)


def _limit_columns(m: torch.Tensor, valid) -> torch.Tensor:
    """CPU path: zero mask columns at or behind ``valid[i]``."""
    m = m.bool()
    if valid is None:
        return m
    return m & (torch.arange(m.shape[1], device=m.device).unsqueeze(0) < valid.to(m.device).unsqueeze(1))


def _compact_indices(m: torch.Tensor, valid, bound: Optional[int] = None):
    """(indices [B, M] int64 zero-filled, counts [B] int64, result width).  The width is the largest count, which
    costs the one host synchronisation of these operators (it decides an output SHAPE) — unless the caller supplies
    an upper ``bound``: then the width is the bound, counts are clamped to it, and nothing synchronises."""
    if m.is_cuda:
        idx, sizes = _ext.mask_to_indices(m, valid)
    else:
        mm = _limit_columns(m, valid)
        sizes = mm.sum(dim=1, dtype=torch.int64)
        order = torch.argsort((~mm).to(torch.int8), dim=1, stable=True)
        keep = torch.arange(m.shape[1]).unsqueeze(0) < sizes.unsqueeze(1)
        idx = torch.where(keep, order, torch.zeros_like(order))
    if bound is not None:
        longest = max(0, min(int(bound), m.shape[1]))
        return idx, sizes.clamp(max=longest), longest
    longest = int(sizes.max().item()) if sizes.numel() > 0 else 0
    return idx, sizes, longest


def _gather(data_t: torch.Tensor, idx: torch.Tensor, sizes: torch.Tensor, longest: int) -> torch.Tensor:
    """out[i, j] = data_t[i, idx[i, j]] (j < sizes[i]), zeros elsewhere; data_t is [B, M, *inner]."""
    b = data_t.shape[0]
    if data_t.is_cuda:
        if data_t.dtype in _AUTOGRAD_DTYPES:
            return BatchedIndexingAccess.apply(data_t, idx[:, :longest].contiguous(), sizes, 0.0)
        out = torch.zeros((b, longest) + tuple(data_t.shape[2:]), dtype=data_t.dtype, device=data_t.device)
        _ext.gather_rows(data_t.contiguous(), idx, sizes, longest, out)
        return out
    out = torch.zeros((b, longest) + tuple(data_t.shape[2:]), dtype=data_t.dtype, device=data_t.device)
    if longest == 0 or b == 0:
        return out
    keep = torch.arange(longest).unsqueeze(0) < sizes.unsqueeze(1)
    rows = torch.arange(b).unsqueeze(1).expand(b, longest)[keep]
    out = out.index_put((rows, torch.arange(longest).unsqueeze(0).expand(b, longest)[keep]),
                        data_t[rows, idx[:, :longest][keep]])
    return out


def batched_bool_indexing(input_data: Union[RaggedBatch, torch.Tensor],
                          input_mask: Union[RaggedBatch, torch.Tensor], *,
                          max_sample_size: Optional[int] = None) -> RaggedBatch:
    """Per-sample boolean indexing along the non-uniform dimension (``dim == 1`` for plain tensors): sample ``i`` of
    the result holds, in order, the entries of ``input_data[i]`` whose mask entry is True.  The result is a
    RaggedBatch with the per-sample True counts as sample sizes and zeros in the padding.  When exactly one
    argument is a RaggedBatch its sample sizes also bound the other argument; with two tensors every column is
    valid.  Several batch dimensions are supported when both arguments are RaggedBatch instances.

    ``max_sample_size`` (extension): an upper bound of the per-sample True counts known to the caller.  The result
    then has exactly that width and the call does not synchronise with the host (usable inside a captured graph);
    without it the width is the largest count, read back from the device as in the reference
    (batched_bool_indexing.py:198).  Entries beyond the bound are dropped."""
    _check_pair(input_data, input_mask)
    d_rb = isinstance(input_data, RaggedBatch)
    batch_shape = input_data.batch_shape if d_rb else torch.Size([input_data.shape[0]])
    multi = d_rb and input_data.num_batch_dims > 1
    m, valid = _flat_mask_and_valid(input_mask, input_data)
    if d_rb:
        orig_dim = input_data.non_uniform_dim
        if multi:
            input_data = input_data.flatten_batch_dims()
        data_t = input_data.get_non_uniform_dimension_transposed_to(1).tensor
    else:
        data_t = input_data
    idx, sizes, longest = _compact_indices(m, valid, max_sample_size)
    out = RaggedBatch(_gather(data_t, idx, sizes, longest), sample_sizes=sizes, non_uniform_dim=1)
    if d_rb:
        if multi:
            out = out.reshape_batch_dims(tuple(batch_shape))
        out = out.get_non_uniform_dimension_transposed_to(orig_dim)
    return out


def batched_bool_indexing_write(to_write: RaggedBatch, output_mask: Union[RaggedBatch, torch.Tensor],
                                to_write_into: Union[RaggedBatch, torch.Tensor]) -> Union[RaggedBatch, torch.Tensor]:
    """Inverse of :func:`batched_bool_indexing`: on a copy of ``to_write_into``, the k-th True position of
    ``output_mask[i]`` receives the k-th valid entry of ``to_write[i]``; everything else is unchanged.  Returns the
    type of ``to_write_into``."""
    assert isinstance(to_write, RaggedBatch), "to_write must be a RaggedBatch"
    _check_pair(to_write_into, output_mask)
    into_rb = isinstance(to_write_into, RaggedBatch)
    batch_shape = to_write.batch_shape
    multi = to_write.num_batch_dims > 1
    assert to_write.dim() == to_write_into.dim(), "to_write and to_write_into must have the same number of dimensions"
    m, valid = _flat_mask_and_valid(output_mask, to_write_into)
    if multi:
        to_write = to_write.flatten_batch_dims()
    to_write = to_write.get_non_uniform_dimension_transposed_to(1)
    if into_rb:
        orig_dim = to_write_into.non_uniform_dim
        if multi:
            to_write_into = to_write_into.flatten_batch_dims()
        to_write_into = to_write_into.get_non_uniform_dimension_transposed_to(1)
        dst = to_write_into.tensor
    else:
        dst = to_write_into
    src = to_write.tensor
    if dst.is_cuda:
        idx, sizes = _ext.mask_to_indices(m, valid)
        counts = torch.minimum(sizes, to_write.sample_sizes.reshape(-1).to(torch.int64))
        width = src.shape[1]
        if src.dtype in _AUTOGRAD_DTYPES and src.dtype == dst.dtype:
            res = BatchedInverseIndexingAccessInsert.apply(src, idx[:, :width].contiguous(), counts, dst)
        else:
            res = dst.clone()
            _ext.scatter_rows(src.to(dst.dtype).contiguous(), idx, counts, width, res)
    else:
        mm = _limit_columns(m, valid)
        res = dst.clone()
        res[mm] = src[to_write.mask]
    if into_rb:
        res = to_write_into.create_with_sample_sizes_like_self(res, 1)
        if multi:
            res = res.reshape_batch_dims(tuple(batch_shape))
        res = res.get_non_uniform_dimension_transposed_to(orig_dim)
    return res


def get_compact_lists(mask: torch.Tensor, data: Sequence[Union[torch.Tensor, Any]], *,
                      max_sample_size: Optional[int] = None) -> List[Union[RaggedBatch, Any]]:
    """Left-compact every tensor of ``data`` (each ``[B, M, ...]``) along ``dim == 1`` by the 2-D ``mask`` into a
    RaggedBatch of width ``max(mask.sum(1))``; non-tensor items pass through unchanged.  The compaction indices
    are computed once and shared by all tensors.  ``max_sample_size`` (extension): see :func:`batched_bool_indexing`."""
    m = mask.bool() if mask.dtype != torch.bool else mask
    idx, sizes, longest = _compact_indices(m, None, max_sample_size)
    out: List[Any] = []
    for el in data:
        if isinstance(el, torch.Tensor):
            t = el if el.dim() >= 2 else el.unsqueeze(1)
            if t.device != idx.device:
                g = _gather(t, idx.to(t.device), sizes.to(t.device), longest)
                out.append(RaggedBatch(g, sample_sizes=sizes.to(t.device)))
            else:
                out.append(RaggedBatch(_gather(t, idx, sizes, longest), sample_sizes=sizes))
        else:
            out.append(el)
    return out


def get_compact_from_named_tuple(mask: torch.Tensor, data: NamedTuple, *,
                                 max_sample_size: Optional[int] = None) -> NamedTuple:
    """:func:`get_compact_lists` for a named tuple; the result has the same named-tuple type."""
    return type(data)(*get_compact_lists(mask, data, max_sample_size=max_sample_size))


def get_indices_from_mask(mask: Union[torch.Tensor, RaggedBatch], *,
                          max_sample_size: Optional[int] = None) -> RaggedBatch:
    """Per sample, the positions of the True mask entries (int64, in order) as a RaggedBatch.  2-D masks only
    (one batch dimension)."""
    valid = None
    if isinstance(mask, RaggedBatch):
        assert mask.num_batch_dims == 1, "Only RaggedBatch instances with a single batch dimension are supported"
        valid = mask.sample_sizes
        mask = mask.tensor
    assert mask.ndim == 2, "Only 2D masks (batch_size, num_elements) are supported"
    idx, sizes, longest = _compact_indices(mask, valid, max_sample_size)
    return RaggedBatch(idx[:, :longest].contiguous(), sample_sizes=sizes)
