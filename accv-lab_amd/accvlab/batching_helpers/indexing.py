"""Ragged gather / scatter / index-pair mapping with autograd, and mask-from-indices.

Public functions and their autograd.Function classes keep the names and semantics of the reference
(batched_indexing_ops.py:22-165,168-455; batched_index_mapping_op.py:22-219; batched_mask_from_indices.py:20-61).
All tensor work goes through ``batched_indexing_access_cuda`` (this package's binding of libaccv_hip.so).
"""
from __future__ import annotations

from typing import Any, Optional, Union

import torch
from torch.autograd.function import once_differentiable

from . import batched_indexing_access_cuda as _ext
from .ragged import RaggedBatch


class BatchedIndexingAccess(torch.autograd.Function):
    """forward: gather ``data[i, idx[i, j]]``; backward: scatter-ADD of the gradient (duplicates accumulate)."""

    @staticmethod
    def forward(ctx: Any, input_data, input_indices, input_nums_indices, fill_value: float = 0.0):
        input_data, input_indices = input_data.contiguous(), input_indices.contiguous()
        input_nums_indices = input_nums_indices.contiguous()
        out = _ext.forward(input_data, input_indices, input_nums_indices, fill_value)
        ctx.save_for_backward(input_indices, input_nums_indices)
        ctx.input_num_targets = input_data.shape[input_nums_indices.dim()]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx: Any, grad):
        if grad is None:
            return None, None, None, None
        indices, counts = ctx.saved_tensors
        g = _ext.backward_new_tensor(grad.contiguous(), indices, counts, ctx.input_num_targets, 0.0,
                                     backward_accumulate=True)
        return g, None, None, None


class BatchedInverseIndexingAccessNewTensor(torch.autograd.Function):
    """forward: scatter into a fresh constant tensor; backward: gather."""

    @staticmethod
    def forward(ctx: Any, input, output_indices, output_nums_indices, output_num_targets, fill_value: float = 0.0):
        input, output_indices = input.contiguous(), output_indices.contiguous()
        output_nums_indices = output_nums_indices.contiguous()
        out = _ext.backward_new_tensor(input, output_indices, output_nums_indices, output_num_targets, fill_value,
                                       backward_accumulate=False)
        ctx.save_for_backward(output_indices, output_nums_indices)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx: Any, grad):
        if grad is None:
            return None, None, None, None, None
        indices, counts = ctx.saved_tensors
        return _ext.forward(grad.contiguous(), indices, counts, 0.0), None, None, None, None


class BatchedInverseIndexingAccessInsert(torch.autograd.Function):
    """forward: scatter into a clone of ``to_fill_into``; backward: gather for the source and
    zero-at-indices for the destination."""

    @staticmethod
    def forward(ctx: Any, to_fill, output_indices, output_nums_indices, to_fill_into):
        to_fill, to_fill_into = to_fill.contiguous(), to_fill_into.contiguous()
        output_indices, output_nums_indices = output_indices.contiguous(), output_nums_indices.contiguous()
        out = _ext.backward_insert(to_fill, output_indices, output_nums_indices, to_fill_into)
        ctx.save_for_backward(output_indices, output_nums_indices)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx: Any, grad):
        if grad is None:
            return None, None, None, None
        indices, counts = ctx.saved_tensors
        grad = grad.contiguous()
        g_src = _ext.forward(grad, indices, counts, 0.0)
        g_dst = _ext.backward_insert_const(0.0, indices, counts, grad)
        return g_src, None, None, g_dst


class BatchedIndexMapping(torch.autograd.Function):
    """forward: ``out[i, dst_idx[i,j]] = src[i, src_idx[i,j]]`` on a clone; backward: accumulate back along the
    same pairs (sources may repeat) and zero the overwritten slots of the destination gradient."""

    @staticmethod
    def forward(ctx: Any, input_data, input_indices, output_indices, nums_indices, to_insert_into):
        input_data, to_insert_into = input_data.contiguous(), to_insert_into.contiguous()
        input_indices, output_indices = input_indices.contiguous(), output_indices.contiguous()
        nums_indices = nums_indices.contiguous()
        out = _ext.map_values_by_index_pairs(input_data, input_indices, output_indices, nums_indices, to_insert_into,
                                             backward_accumulate=False)
        ctx.save_for_backward(input_indices, output_indices, nums_indices)
        ctx.num_batch_dims = nums_indices.dim()
        ctx.input_max_sample_size = input_data.shape[ctx.num_batch_dims]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx: Any, grad):
        if grad is None:
            return None, None, None, None, None
        src_idx, dst_idx, counts = ctx.saved_tensors
        grad = grad.contiguous()
        shape = list(grad.shape)
        shape[ctx.num_batch_dims] = ctx.input_max_sample_size
        zeros = torch.zeros(shape, dtype=grad.dtype, device=grad.device)
        g_src = _ext.map_values_by_index_pairs(grad, dst_idx, src_idx, counts, zeros, backward_accumulate=True)
        g_dst = _ext.backward_insert_const(0.0, dst_idx, counts, grad)
        return g_src, None, None, None, g_dst


def _run(fn, *args):
    """``fn.apply(*args)``, or — when no argument needs a gradient — the forward alone: building an autograd node costs
    several microseconds per call on a launch-bound path and records nothing useful then."""
    if torch.is_grad_enabled():
        for a in args:
            if isinstance(a, torch.Tensor) and a.requires_grad:
                return fn.apply(*args)
    out = fn.forward(_NoCtx(), *args)
    # the binding allocates its result with requires_grad = input.requires_grad (as the reference's torch::full does);
    # under torch.no_grad() Function.apply would hand back a tensor that does NOT require grad -> same here
    if isinstance(out, torch.Tensor) and out.requires_grad:
        out.requires_grad_(False)
    return out


class _NoCtx:
    """Stand-in for the autograd context on the no-grad fast path."""

    def save_for_backward(self, *tensors):
        pass


def _resolve_dim(indices: RaggedBatch, dim_to_index_in: Optional[int], what: str):
    if dim_to_index_in is None:
        dim_to_index_in = indices.non_uniform_dim
    assert dim_to_index_in >= indices.num_batch_dims, f"Cannot index in a batch dimension of the {what}"
    return dim_to_index_in, indices.num_batch_dims != dim_to_index_in


def batched_indexing_access(input_data: Union[RaggedBatch, torch.Tensor], input_indices: RaggedBatch,
                            filler_value: float = 0.0, dim_to_index_in: Optional[int] = None) -> RaggedBatch:
    """Ragged gather: ``out[i, j] == input_data[i, input_indices[i, j]]`` for ``j < input_indices.sample_sizes[i]``
    along ``dim_to_index_in`` (default: the indices' non-uniform dimension); ``filler_value`` in the padding.
    Differentiable w.r.t. ``input_data`` (repeated indices accumulate).  Returns a RaggedBatch sharing the
    indices' sample sizes.  GPU only."""
    data = input_data.tensor if isinstance(input_data, RaggedBatch) else input_data
    dim, swap = _resolve_dim(input_indices, dim_to_index_in, "input indices")
    nb = input_indices.num_batch_dims
    if swap:
        data = data.transpose(nb, dim)
    out = _run(BatchedIndexingAccess, data, input_indices.tensor, input_indices.sample_sizes, filler_value)
    if swap:
        out = out.transpose(nb, dim)
    return input_indices.create_with_sample_sizes_like_self(out, dim)


def batched_inverse_indexing_access(input_data: Union[RaggedBatch, torch.Tensor], output_indices: RaggedBatch,
                                    output_num_targets: int, filler_value: float = 0.0,
                                    dim_to_index_in: Optional[int] = None) -> torch.Tensor:
    """Ragged scatter into a fresh tensor: ``out[i, output_indices[i, j]] == input_data[i, j]``, ``filler_value``
    elsewhere; ``out.shape[dim_to_index_in] == output_num_targets``.  Indices of one sample must be unique.
    Differentiable w.r.t. ``input_data``.  GPU only."""
    data = input_data.tensor if isinstance(input_data, RaggedBatch) else input_data
    dim, swap = _resolve_dim(output_indices, dim_to_index_in, "output indices")
    nb = output_indices.num_batch_dims
    if swap:
        data = data.transpose(nb, dim)
    out = _run(BatchedInverseIndexingAccessNewTensor, data, output_indices.tensor, output_indices.sample_sizes,
                                                      output_num_targets, filler_value)
    return out.transpose(nb, dim) if swap else out


def batched_indexing_write(to_write: Union[RaggedBatch, torch.Tensor], output_indices: RaggedBatch,
                           to_write_into: Union[RaggedBatch, torch.Tensor],
                           dim_to_index_in: Optional[int] = None) -> Union[RaggedBatch, torch.Tensor]:
    """Ragged scatter into a copy of ``to_write_into``: ``out[i, output_indices[i, j]] == to_write[i, j]``, all other
    entries as in ``to_write_into``.  Returns the type of ``to_write_into``.  Differentiable w.r.t. both data
    arguments.  GPU only."""
    dim, swap = _resolve_dim(output_indices, dim_to_index_in, "output indices")
    nb = output_indices.num_batch_dims
    src = to_write.tensor if isinstance(to_write, RaggedBatch) else to_write
    dst_is_rb = isinstance(to_write_into, RaggedBatch)
    dst = to_write_into.tensor if dst_is_rb else to_write_into
    if swap:
        src, dst = src.transpose(nb, dim), dst.transpose(nb, dim)
    out = _run(BatchedInverseIndexingAccessInsert, src, output_indices.tensor, output_indices.sample_sizes, dst)
    if swap:
        out = out.transpose(nb, dim)
    return to_write_into.create_with_sample_sizes_like_self(out) if dst_is_rb else out


def batched_index_mapping(source_data: Union[torch.Tensor, RaggedBatch], source_indices: RaggedBatch,
                          target_indices: RaggedBatch,
                          target_data: Union[torch.Tensor, RaggedBatch]) -> Union[torch.Tensor, RaggedBatch]:
    """``out[i, target_indices[i, j]] = source_data[i, source_indices[i, j]]`` for every valid pair ``j`` on a copy of
    ``target_data`` (returned with its type).  Source and target indices share sample sizes; target indices of one
    sample must be unique, source indices may repeat.  Differentiable w.r.t. both data arguments.  GPU only."""
    nb = target_indices.non_uniform_dim
    assert target_indices.dim() == nb + 1 and source_indices.dim() == nb + 1, \
        "Indices must have exactly one dimension in addition to the batch dimensions"
    assert target_indices.shape[:nb] == source_indices.shape[:nb], "Batch shape mismatch"
    assert target_indices.shape[nb] == source_indices.shape[nb], "Maximum number of indices mismatch"
    tgt_is_rb = isinstance(target_data, RaggedBatch)
    if tgt_is_rb:
        tgt_dim = target_data.non_uniform_dim
        target_data = target_data.get_non_uniform_dimension_transposed_to(nb)
        tgt = target_data.tensor
    else:
        tgt = target_data
    if isinstance(source_data, RaggedBatch):
        source_data = source_data.get_non_uniform_dimension_transposed_to(nb).tensor
    out = _run(BatchedIndexMapping, source_data, source_indices.tensor, target_indices.tensor,
                                    source_indices.sample_sizes, tgt)
    if tgt_is_rb:
        out = target_data.create_with_sample_sizes_like_self(out, nb)
        out = out.get_non_uniform_dimension_transposed_to(tgt_dim)
    return out


def get_mask_from_indices(mask_num_targets: int, indices: RaggedBatch) -> torch.Tensor:
    """bool ``[*batch, mask_num_targets]`` with ``mask[i, indices[i, :indices.sample_sizes[i]]] = True``.  GPU only."""
    return _ext.get_mask_from_indices(indices.tensor.contiguous(), indices.sample_sizes.contiguous(), mask_num_targets)
