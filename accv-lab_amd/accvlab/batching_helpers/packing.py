"""List-of-tensors -> RaggedBatch packing and masked reductions.

Public contract as in the reference (batched_processing_py.py:23-174 reductions/helpers, :275-574
combine_data).  combine_data differs in HOW it builds the padded tensor: the reference issues one python
slice-assignment per sample (:416-423); here the samples are concatenated once, moved to the target device
in ONE transfer and unpacked by a single kernel (accv_ragged_pack) — or, on CPU / when gradients must flow,
by one vectorised index_put.
"""
from __future__ import annotations

from collections.abc import Sequence as _Seq
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import batched_indexing_access_cuda as _ext
from .ragged import RaggedBatch, remember_host_sizes

try:  # C++ per-sample loops (built by `make -C accv-lab_amd/csrc_host`); the python code below is the fallback
    from . import _bh_host as _bh
except ImportError:  # pragma: no cover
    _bh = None


# ------------------------------------------------------------------------------------------ reductions
def sum_over_targets(data: RaggedBatch) -> torch.Tensor:
    """Sum of the valid entries along the non-uniform dimension (one value per sample and data element)."""
    masked = data.with_padded_set_to(0.0)
    return torch.sum(masked.tensor, dim=data.non_uniform_dim, dtype=masked.tensor.dtype)


def average_over_targets(data: RaggedBatch, nans_to_zero: bool = True) -> torch.Tensor:
    """Mean of the valid entries along the non-uniform dimension.  Empty samples give NaN, replaced by 0 when
    ``nans_to_zero`` (default)."""
    nb = data.num_batch_dims
    data = data.get_non_uniform_dimension_transposed_to(nb)
    total = torch.sum(data.with_padded_set_to(0.0).tensor, dim=nb, dtype=data.dtype)
    sizes = data.sample_sizes.reshape(tuple(data.sample_sizes.shape) + (1,) * (total.dim() - nb))
    res = total / sizes
    if nans_to_zero:
        res = torch.nan_to_num(res, nan=0.0, posinf=0.0, neginf=0.0)
    return res


def apply_mask_to_tensor(data: torch.Tensor, mask: torch.Tensor, value_to_set: float = 0.0) -> torch.Tensor:
    """Copy of ``data`` with ``value_to_set`` wherever ``mask`` is False; ``mask`` covers the leading dimensions of
    ``data`` and is broadcast over the rest."""
    if mask.numel() == 0:
        return data
    extra = data.dim() - mask.dim()
    if extra > 0:
        mask = mask.reshape(tuple(mask.shape) + (1,) * extra)
    return data.masked_fill(torch.logical_not(mask.expand(data.shape)), value_to_set)


def squeeze_except_batch_and_sample(data: Union[torch.Tensor, RaggedBatch]) -> Union[torch.Tensor, RaggedBatch]:
    """Squeeze all size-1 dimensions except the batch dimension(s) and the non-uniform ("sample") dimension.  Plain
    tensors are taken as ``[batch, sample, ...]``.  For a RaggedBatch the non-uniform dimension index shifts by the
    number of squeezed dimensions in front of it."""
    if not isinstance(data, RaggedBatch):
        return torch.squeeze(data, tuple(range(2, data.dim())))
    nb, nu = data.num_batch_dims, data.non_uniform_dim
    dims = tuple(range(nb, nu)) + tuple(range(nu + 1, data.dim()))
    dropped_before = sum(1 for s in data.shape[nb:nu] if s == 1)
    return data.create_with_sample_sizes_like_self(torch.squeeze(data.tensor, dims), non_uniform_dim=nu - dropped_before)


# ------------------------------------------------------------------------------------------ combine_data
def _is_seq(x) -> bool:
    return isinstance(x, _Seq) and not isinstance(x, str)


def _valid_len(t: torch.Tensor) -> int:
    return min(int(t.shape[0]), t.numel()) if t.dim() > 0 else 0


def _walk_flat(data, out: List[torch.Tensor]) -> None:
    if isinstance(data, torch.Tensor):
        out.append(data)
    elif _is_seq(data):
        for el in data:
            _walk_flat(el, out)
    else:
        raise AssertionError(f"The data to combine must be a tensor or a (nested) sequence of tensors. Got {type(data)}.")


def _nested_shape(data, level=0) -> List[int]:
    assert _is_seq(data), "`data` must be a sequence"
    first = data[0]
    if isinstance(first, torch.Tensor):
        for item in data[1:]:
            if not isinstance(item, torch.Tensor):
                raise AssertionError("The data to combine must be a tensor or a (nested) sequence of tensors. "
                                     f"Got {type(item)} contained in the sequence at level {level}.")
        return [len(data)]
    sub = _nested_shape(first, level + 1)
    for item in data[1:]:
        if not _is_seq(item):
            raise AssertionError("The data to combine must be a tensor or a (nested) sequence of tensors. "
                                 f"Got {type(item)} contained in the sequence at level {level + 1}.")
        other = _nested_shape(item, level + 1)
        if other != sub:
            raise AssertionError(f"Inconsistent sequence length structure at level {level + 1}. "
                                 f"Expected shape {sub}, got {other}")
    return [len(data)] + sub


def _build_padded(leaves: List[torch.Tensor], lens: List[int], width: int, proto: torch.Tensor,
                  device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(padded [B, width, *inner] on `device`, int64 sizes on cpu)."""
    b = len(leaves)
    inner = tuple(proto.shape[1:])
    sizes_cpu = torch.tensor(lens, dtype=torch.int64)
    device = torch.device(device)
    total = int(sum(lens))
    if total == 0 or width == 0:
        return torch.zeros((b, width) + inner, dtype=proto.dtype, device=device), sizes_cpu

    def norm(t, n):
        if n == 0:
            return proto.new_zeros((0,) + inner)
        if t.shape[0] != n:
            t = t[:n]
        return t if t.dtype == proto.dtype else t.to(dtype=proto.dtype)

    needs_grad = any(t.requires_grad for t in leaves)
    if device.type == "cuda" and not needs_grad:
        parts = [norm(t, n) for t, n in zip(leaves, lens) if n > 0]
        if any(p.device != parts[0].device for p in parts):
            parts = [p.to(device) for p in parts]
        flat = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        if flat.device != device:
            flat = flat.to(device, non_blocking=True)
        offsets_cpu = torch.cumsum(sizes_cpu, 0) - sizes_cpu
        meta = torch.stack([offsets_cpu, sizes_cpu]).to(device, non_blocking=True)
        return _ext.pack_rows(flat.contiguous(), meta[0], meta[1], width), sizes_cpu
    # CPU target, or gradients must flow: cat + ONE index_copy (both differentiable); indices built with numpy
    parts = [norm(t, n) for t, n in zip(leaves, lens) if n > 0]
    if any(q.device != device for q in parts):
        parts = [q.to(device) for q in parts]
    flat = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    ln = np.asarray(lens, dtype=np.int64)
    starts = np.cumsum(ln) - ln
    dest = np.repeat(np.arange(b, dtype=np.int64) * width - starts, ln) + np.arange(total, dtype=np.int64)
    padded = torch.zeros((b * width,) + inner, dtype=proto.dtype, device=device)
    padded = padded.index_copy(0, torch.from_numpy(dest).to(device), flat)
    return padded.view((b, width) + inner), sizes_cpu


_DIRECT_PADDED_BYTES = 1 << 20


def _fast_pack(data_list, device):
    """combine_data (flatten mode) through the C++ loops of _bh_host; None = not a plain case, use the python path.
    Returns (padded on the target device, int64 sizes on the target device)."""
    target = None if device is None else torch.device(device)
    if target is not None and target.type == "cuda" and target.index is None:
        target = torch.device("cuda", torch.cuda.current_device())   # "cuda" == the current device
    if target is None or target.type == "cpu":
        fast = _bh.pack_cpu(data_list, False)     # declines unless every sample is a plain CPU tensor
        if fast is not None:
            return fast
        if target is not None:
            return None
    if target is not None and target.type != "cuda":
        return None
    first = data_list[0]
    while _is_seq(first) and len(first):
        first = first[0]
    if not isinstance(first, torch.Tensor):
        return None
    if first.device.type == "cpu":
        if target is None:
            return None
        # CPU samples -> GPU: small batches travel PADDED (pinned, one asynchronous copy, no kernel at all)
        # (bigger batches: flat transfer + pack kernel on the python path, so that the padding never crosses the link)
        small = _bh.pack_cpu(data_list, True, _DIRECT_PADDED_BYTES)
        if small is None:
            return None
        return small[0].to(target, non_blocking=True), \
            remember_host_sizes(small[1].to(target, non_blocking=True), small[1].tolist())
    if first.device.type != "cuda" or (target is not None and target != first.device):
        return None
    got = _bh.cat_leaves(data_list, True)         # GPU samples: trim/check loop + cat in C++, then the pack kernel
    if got is None:
        return None
    flat, sizes_cpu, meta, width = got
    if flat.numel() == 0 or width == 0:
        return None
    meta_dev = meta.to(flat.device, non_blocking=True)
    return _ext.pack_rows(flat, meta_dev[0], meta_dev[1], width), remember_host_sizes(meta_dev[1], sizes_cpu.tolist())


def combine_data(data_list: Sequence[Union[Sequence, torch.Tensor]], other_with_same_sample_sizes: RaggedBatch = None,
                 device: Optional[Union[torch.device, str]] = None, flatten_batch_dims: bool = True) -> RaggedBatch:
    """Pack a (nested) sequence of per-sample tensors ``(n_i, *d)`` into one RaggedBatch ``[*batch, max n_i, *d]``.

    ``flatten_batch_dims=True`` (default): any nesting is flattened depth-first into ONE batch dimension.
    ``flatten_batch_dims=False``: every nesting level becomes a batch dimension (all lists of a level must have the
    same length).  ``other_with_same_sample_sizes`` lets the result share that batch's mask and sample sizes (they are
    trusted, not checked).  ``device`` defaults to the device of the first (non-empty) sample.
    """
    assert _is_seq(data_list), "`data_list` must be a sequence"
    assert len(data_list) > 0, "`data_list` must not be empty"
    share = other_with_same_sample_sizes

    if flatten_batch_dims:
        fast = _fast_pack(data_list, device) if _bh is not None else None
        if fast is not None:
            padded, sizes = fast
            if share is None:
                return RaggedBatch(padded, sample_sizes=sizes)
            assert padded.shape[0] == share.sample_sizes.shape[0], \
                "Number of samples does not match `other_with_same_sample_sizes`"
            assert tuple(padded.shape[:2]) == tuple(share.mask.shape), \
                "Needed mask dimension does not match `other_with_same_sample_sizes`"
            return share.create_with_sample_sizes_like_self(padded, non_uniform_dim=1, device=padded.device)
        leaves: List[torch.Tensor] = []
        _walk_flat(data_list, leaves)
        if not leaves:
            return RaggedBatch.Empty(2, 1, device=device)
        width = max(int(t.shape[0]) for t in leaves)
        proto = next((t for t in leaves if t.numel() > 0), leaves[0])
        if device is None:
            device = proto.device
        lens = [_valid_len(t) for t in leaves]
        padded, sizes_cpu = _build_padded(leaves, lens, width, proto, device)
        if share is None:
            return RaggedBatch(padded, sample_sizes=remember_host_sizes(sizes_cpu.to(device), lens))
        assert len(leaves) == share.sample_sizes.shape[0], "Number of samples does not match `other_with_same_sample_sizes`"
        assert (len(leaves), width) == tuple(share.mask.shape), \
            "Needed mask dimension does not match `other_with_same_sample_sizes`"
        return share.create_with_sample_sizes_like_self(padded, non_uniform_dim=1, device=device)

    batch_shape = _nested_shape(data_list)
    nb = len(batch_shape)
    leaves = []
    _walk_flat(data_list, leaves)
    width = share.mask.shape[-1] if share is not None else max([int(t.shape[0]) for t in leaves] + [0])
    proto = leaves[0] if leaves else None
    if proto is None or width == 0:
        dev = torch.device("cpu") if device is None else device
        return RaggedBatch(torch.empty((*batch_shape, 0), dtype=torch.float32, device=dev),
                           torch.empty((*batch_shape, 0), dtype=torch.bool, device=dev),
                           torch.zeros(batch_shape, dtype=torch.int64, device=dev), non_uniform_dim=nb)
    if device is None:
        device = proto.device
    lens = [_valid_len(t) for t in leaves]
    padded, sizes_cpu = _build_padded(leaves, lens, width, proto, device)
    padded = padded.reshape(*batch_shape, *padded.shape[1:])
    if share is None:
        return RaggedBatch(padded, sample_sizes=remember_host_sizes(sizes_cpu.reshape(batch_shape).to(device), lens),
                           non_uniform_dim=nb)
    assert tuple(share.sample_sizes.shape) == tuple(batch_shape), "Sample sizes shape does not match required batch shape"
    return share.create_with_sample_sizes_like_self(padded, non_uniform_dim=nb, device=device)
