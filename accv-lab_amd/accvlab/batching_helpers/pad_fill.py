"""SetPaddedTo — differentiable in-place pad fill (reference: data_format/set_padded_to.py:20-43)."""
from __future__ import annotations

import torch

from . import batched_indexing_access_cpu as _cpu
from . import batched_indexing_access_cuda as _gpu


class SetPaddedTo(torch.autograd.Function):
    """``data[i, j, ...] = value`` for ``j >= sample_sizes[i]``.  Operates in place on ``data`` when it is
    contiguous (otherwise on a contiguous copy) and returns the filled tensor.  Backward: the incoming
    gradient with its padding zeroed."""

    @staticmethod
    def forward(ctx, data: torch.Tensor, sample_sizes: torch.Tensor, value_to_set):
        ctx.save_for_backward(sample_sizes)
        data = data.contiguous()
        sizes = sample_sizes.to(dtype=torch.int64).contiguous()
        if data.device.type == "cuda":
            _gpu.set_ragged_batch_padded_to_filler_value_in_place(data, sizes, value_to_set)
        else:
            _cpu.set_ragged_batch_padded_to_filler_value_in_place(data, sizes, value_to_set)
        return data

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output is None:
            return None, None, None
        grad = grad_output.clone()
        SetPaddedTo.apply(grad, ctx.saved_tensors[0], 0.0)
        return grad, None, None
