"""Drop-in for the reference's CPU extension module ``accvlab.batching_helpers.batched_indexing_access_cpu``
(cpp_impl/batched_indexing_access_cpu.cpp:43-47): exports the pad-fill op only.

The reference loops over samples under at::parallel_for (batched_indexing_access_cpu_impl.cpp:27-44); for
CPU tensors this build uses one vectorised torch ``masked_fill_`` — this is the CPU *product* path for CPU
tensors (as in the reference), not a fallback for the GPU path.
"""
from __future__ import annotations

import torch


def set_ragged_batch_padded_to_filler_value_in_place(data: torch.Tensor, nums_valid_entries: torch.Tensor,
                                                     filler_value: float) -> None:
    if not data.device.type == "cpu":
        raise RuntimeError("data must be a CPU tensor")
    if not nums_valid_entries.device.type == "cpu":
        raise RuntimeError("nums_valid_entries must be a CPU tensor")
    if not data.is_contiguous():
        raise RuntimeError("data must be contiguous")
    nb = nums_valid_entries.dim()
    if data.numel() == 0:
        return
    if data.dim() < nb + 1:
        raise RuntimeError(f"data must have at least {nb + 1} dimensions")
    if tuple(data.shape[:nb]) != tuple(nums_valid_entries.shape):
        raise RuntimeError("data and nums_valid_entries must have the same size in the batch dimensions")
    width = data.size(nb)
    pad = torch.arange(width).reshape((1,) * nb + (width,)) >= nums_valid_entries.unsqueeze(-1)
    pad = pad.reshape(pad.shape + (1,) * (data.dim() - nb - 1))
    data.masked_fill_(pad, filler_value)
