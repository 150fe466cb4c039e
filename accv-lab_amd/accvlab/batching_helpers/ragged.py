"""RaggedBatch — padded tensor + per-sample sizes (+ lazily derived validity mask).

API-compatible with the reference type (packages/batching_helpers/accvlab/batching_helpers/data_format/
ragged_batch.py:31-1111): same constructor, class methods, properties and methods, same assertion
behaviour.  Layout: ``tensor[*batch_shape, ..., max_sample_size (at non_uniform_dim), ...]``,
``mask[*batch_shape, max_sample_size]`` (bool), ``sample_sizes[*batch_shape]`` (int64).  Valid entries
come first along the non-uniform dimension; whatever sits in the padding is unspecified.

The mask is produced on demand by the pad-fill kernel (GPU: accv_ragged_pad_fill in libaccv_hip.so;
CPU tensors: a torch comparison), exactly where the reference calls SetPaddedTo (ragged_batch.py:220-228).
"""
from __future__ import annotations

import inspect
import math
from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch

from .pad_fill import SetPaddedTo

__all__ = ["RaggedBatch"]


def _as_tuple(shape) -> Tuple[int, ...]:
    return (int(shape),) if isinstance(shape, int) else tuple(int(s) for s in shape)


try:  # C++ per-sample loops (built by `make -C accv-lab_amd/csrc_host`)
    from . import _bh_host as _bh
except ImportError:  # pragma: no cover
    _bh = None


def remember_host_sizes(sizes: torch.Tensor, values) -> torch.Tensor:
    """Attach the host copy of a DEVICE sample-size tensor to it (row-major python ints) so that consumers which need the
    sizes on the host (``split``) do not have to read them back — a device synchronisation.  Only ``combine_data`` does this,
    for size tensors it has just made FROM host values.  The copy is trusted while the tensor's version counter is unchanged,
    so any in-place edit through autograd-visible operations invalidates it.  Writes the version counter does not see —
    ``sizes.data.copy_(...)``, ``set_()``, a kernel that writes through ``data_ptr()``, a graph replay that refills the
    buffer — do not: code that rewrites such a tensor must call :func:`forget_host_sizes` (or build a new RaggedBatch)."""
    try:
        sizes._accv_host_sizes = (list(values), sizes._version)
    except Exception:  # pragma: no cover - tensor subclasses without a __dict__
        pass
    return sizes


def forget_host_sizes(sizes: torch.Tensor) -> torch.Tensor:
    """Drop the host copy attached by ``combine_data`` (see :func:`remember_host_sizes`): the next ``split()`` reads the device
    tensor again."""
    try:
        del sizes._accv_host_sizes
    except AttributeError:
        pass
    return sizes


def host_sizes(sizes: torch.Tensor) -> list:
    """Row-major python ints of ``sizes``; from the host copy ``combine_data`` attached while it is still valid, else ONE
    read-back (which is not cached: a tensor that was read from the device once may be rewritten behind the version
    counter's back, e.g. by a graph replay, and must be read again)."""
    hit = getattr(sizes, "_accv_host_sizes", None)
    if hit is not None and hit[1] == sizes._version and len(hit[0]) == sizes.numel():
        if not (sizes.is_cuda and torch.cuda.is_current_stream_capturing()):
            return hit[0]
    return sizes.reshape(-1).tolist()


class RaggedBatch:
    """Batch whose samples differ in size along one ("non-uniform") dimension.

    Args:
        tensor: padded data, ``(*batch_shape, max_sample_size, *data_shape)`` when the non-uniform dimension
            directly follows the batch dimensions (it may be any later dimension).
        mask: optional bool ``(*batch_shape, max_sample_size)``; True marks valid entries.
        sample_sizes: optional integer ``(*batch_shape,)``.
        non_uniform_dim: defaults to the first dimension after the batch dimensions.

    At least one of ``mask`` / ``sample_sizes`` is required; the number of batch dimensions is taken from
    whichever is given (``sample_sizes`` wins).  If both are given they must agree (not verified).
    ``mask`` and ``sample_sizes`` may be shared between instances: treat them as read-only.
    """

    def __init__(
        self,
        tensor: torch.Tensor,
        mask: Optional[torch.Tensor] = None,
        sample_sizes: Optional[torch.Tensor] = None,
        non_uniform_dim: Optional[int] = None,
    ):
        assert mask is not None or sample_sizes is not None, "At least one of `mask` or `sample_sizes` needs to be set"
        nb = sample_sizes.dim() if sample_sizes is not None else mask.dim() - 1
        assert nb > 0, "Number of batch dimensions needs to be greater than 0"
        assert nb < tensor.dim(), "The number of dimensions of the tensor needs to be at least num_batch_dims + 1"
        if non_uniform_dim is None:
            non_uniform_dim = nb
        assert nb <= non_uniform_dim < tensor.dim(), \
            "Non-uniform dimensions needs to be in the range [num_batch_dims; tensor.dim()["
        if mask is not None:
            assert mask.shape[:nb] == tensor.shape[:nb] and mask.shape[nb] == tensor.shape[non_uniform_dim], (
                "Shape of `tensor` does not match the required shape:\n"
                f"  mask says batch {tuple(mask.shape[:nb])}, max sample size {mask.shape[nb]}\n"
                f"  tensor says batch {tuple(tensor.shape[:nb])}, max sample size {tensor.shape[non_uniform_dim]}")
        if sample_sizes is not None:
            assert sample_sizes.shape[:nb] == tensor.shape[:nb], (
                "Batch shape according to `tensor` does not match the size of `sample_sizes`:\n"
                f"  tensor {tuple(tensor.shape[:nb])} vs sample_sizes {tuple(sample_sizes.shape[:nb])}")
        self._tensor = tensor
        self._mask = mask
        self._sample_sizes = sample_sizes
        self._non_uniform_dim = int(non_uniform_dim)
        self._num_batch_dims = int(nb)
        self._batch_shape = tensor.shape[:nb]
        self._total_entries: Optional[int] = None

    # ------------------------------------------------------------------ alternative constructors
    @classmethod
    def FromOversizeTensor(cls, tensor, mask=None, sample_sizes=None, non_uniform_dim=None) -> "RaggedBatch":
        """Like the constructor, but ``tensor`` (and ``mask``) may be longer than the largest sample along the
        non-uniform dimension; both are narrowed to it.  Needs one device->host read of the maximum size
        (reference: ragged_batch.py:174-218)."""
        if non_uniform_dim is None:
            if sample_sizes is not None:
                non_uniform_dim = sample_sizes.dim()
            elif mask is not None:
                non_uniform_dim = mask.dim() - 1
            else:
                raise ValueError("Either `sample_sizes` or `mask` needs to be set")
        if sample_sizes is None:
            sample_sizes = mask.sum(dim=-1, dtype=torch.int64)
        longest = int(sample_sizes.max().item()) if sample_sizes.numel() > 0 else 0
        tensor = tensor.narrow(non_uniform_dim, 0, longest)
        if mask is not None:
            mask = mask.narrow(mask.dim() - 1, 0, longest)
        return cls(tensor, mask, sample_sizes, non_uniform_dim)

    @classmethod
    def Empty(cls, num_dims: int, non_uniform_dim: int, device, num_batch_dims: Optional[int] = None,
              batch_shape=None) -> "RaggedBatch":
        """Instance with extent 0 along the non-uniform and all data dimensions.  ``batch_shape`` (or
        ``num_batch_dims`` zeros; default one batch dimension of size 0) gives the batch extents."""
        assert num_batch_dims is None or batch_shape is None, \
            "Either num_batch_dims or batch_shape can be provided, but not both"
        if batch_shape is not None:
            batch_shape = _as_tuple(batch_shape)
            assert len(batch_shape) > 0, "Batch shape needs to be a non-empty sequence"
        else:
            nb = 1 if num_batch_dims is None else int(num_batch_dims)
            assert nb > 0, "Number of batch dimensions needs to be greater than 0"
            batch_shape = (0,) * nb
        nb = len(batch_shape)
        assert nb < num_dims, "Number of batch dimensions needs to be less than the total number of dimensions"
        assert nb <= non_uniform_dim < num_dims, "Non-uniform dimension needs to be in the range [num_batch_dims; num_dims["
        tensor = torch.zeros(batch_shape + (0,) * (num_dims - nb), dtype=torch.float32, device=device)
        mask = torch.zeros(batch_shape + (0,), dtype=torch.bool, device=device)
        sizes = torch.zeros(batch_shape, dtype=torch.int64, device=device)
        return cls(tensor, mask, sizes, non_uniform_dim)

    @classmethod
    def FromFullTensor(cls, full_tensor: torch.Tensor, non_uniform_dim: int = 1, num_batch_dims: int = 1) -> "RaggedBatch":
        """Wrap a uniform batch: every sample has the full length of ``non_uniform_dim``."""
        assert num_batch_dims > 0, "Number of batch dimensions needs to be greater than 0"
        assert num_batch_dims <= non_uniform_dim < full_tensor.dim(), \
            f"Non-uniform dimension needs to be in the range [{num_batch_dims}; full_tensor.dim()["
        batch_shape = tuple(full_tensor.shape[:num_batch_dims])
        n = full_tensor.shape[non_uniform_dim]
        mask = torch.ones(batch_shape + (n,), dtype=torch.bool, device=full_tensor.device)
        sizes = torch.full(batch_shape, n, dtype=torch.int64, device=full_tensor.device)
        return cls(full_tensor, mask, sizes, non_uniform_dim)

    # ------------------------------------------------------------------ core attributes
    @property
    def tensor(self) -> torch.Tensor:
        """The padded data tensor (replace it with :meth:`set_tensor`)."""
        return self._tensor

    @property
    def mask(self) -> torch.Tensor:
        """bool ``(*batch_shape, max_sample_size)``; built from ``sample_sizes`` on first use."""
        if self._mask is None:
            t = self._tensor
            n = t.shape[self._non_uniform_dim]
            if t.device.type == "cuda":
                ones = torch.ones(*t.shape[:self._num_batch_dims], n, dtype=torch.bool, device=t.device)
                self._mask = SetPaddedTo.apply(ones, self._sample_sizes, False)   # pad-fill kernel
            elif _bh is not None and hasattr(_bh, "mask_cpu") and self._sample_sizes.device.type == "cpu":
                self._mask = _bh.mask_cpu(self._sample_sizes, n)
            else:
                self._mask = torch.arange(n) < self._sample_sizes.to("cpu").unsqueeze(-1)
        return self._mask

    @property
    def sample_sizes(self) -> torch.Tensor:
        """int64 ``batch_shape``; built from ``mask`` on first use."""
        if self._sample_sizes is None:
            self._sample_sizes = self._mask.sum(dim=-1, dtype=torch.int64)
        return self._sample_sizes

    @property
    def non_uniform_dim(self) -> int:
        return self._non_uniform_dim

    @property
    def num_batch_dims(self) -> int:
        return self._num_batch_dims

    @property
    def batch_shape(self) -> torch.Size:
        return self._batch_shape

    @property
    def total_num_samples_in_batch(self) -> int:
        return int(math.prod(self._batch_shape))

    @property
    def total_num_entries(self) -> int:
        """Sum of all sample sizes (one device->host read, cached)."""
        if self._total_entries is None:
            self._total_entries = int(self.sample_sizes.sum().item())
        return self._total_entries

    @property
    def max_sample_size(self) -> int:
        return int(self._tensor.shape[self._non_uniform_dim])

    # ------------------------------------------------------------------ derived instances
    def as_self_with_cloned_data(self) -> "RaggedBatch":
        """Copy with a cloned data tensor; mask and sizes are shared."""
        return RaggedBatch(self._tensor.clone(), self._mask, self._sample_sizes, self._non_uniform_dim)

    def create_with_sample_sizes_like_self(self, tensor: torch.Tensor, non_uniform_dim: Optional[int] = None,
                                           device=None) -> "RaggedBatch":
        """New instance around ``tensor`` sharing this batch's sizes/mask.  ``tensor`` must agree in the batch
        dimensions and in the extent of its non-uniform dimension; everything else may differ."""
        if non_uniform_dim is None:
            non_uniform_dim = self._non_uniform_dim
        elif non_uniform_dim < 0:
            non_uniform_dim += tensor.dim()
        nb = self._num_batch_dims
        assert nb <= non_uniform_dim < tensor.dim(), f"Non-uniform dimension needs to be in the range [{nb}; {tensor.dim()}["
        assert tensor.shape[:nb] == self.batch_shape, (
            f"Batch shape of tensor does not match required batch shape:\n  Expected batch shape: "
            f"{self.batch_shape}\n  Got batch shape: {tensor.shape[:nb]}")
        assert tensor.shape[non_uniform_dim] == self.max_sample_size, (
            "Non-uniform dimension size of tensor does not match required non-uniform dimension size:\n"
            f"  Expected non-uniform dimension size: {self.max_sample_size}\n"
            f"  Got non-uniform dimension size: {tensor.shape[non_uniform_dim]}")
        if device is None:
            device = tensor.device
        else:
            tensor = tensor.to(device=device)
        # share what has been materialised so far; a missing mask / size tensor stays lazy in the new instance too
        # (the reference forces the mask here, ragged_batch.py:455-457: an extra fill kernel per derived batch)
        mask = self._mask.to(device=device) if self._mask is not None else None
        sizes = self._sample_sizes.to(device=device) if self._sample_sizes is not None else None
        out = RaggedBatch(tensor, mask, sizes, non_uniform_dim)
        out._total_entries = self._total_entries
        return out

    def get_non_uniform_dimension_transposed_to(self, dim: int) -> "RaggedBatch":
        """Swap the non-uniform dimension with data dimension ``dim`` (a view); ``self`` if already there."""
        assert self._num_batch_dims <= dim < self._tensor.dim(), \
            f"Non-uniform dimensions needs to be in the range [{self._num_batch_dims}; tensor.dim()["
        if dim == self._non_uniform_dim:
            return self
        return self.create_with_sample_sizes_like_self(self._tensor.transpose(self._non_uniform_dim, dim), dim)

    def get_existence_weights(self, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """1.0 where an entry is valid, 0.0 in the padding, shaped like ``tensor``."""
        view = list(self._batch_shape) + [1] * (self._tensor.dim() - self._num_batch_dims)
        view[self._non_uniform_dim] = self.max_sample_size
        w = self.mask.to(dtype=dtype).reshape(view)
        return w.expand(self._tensor.shape).contiguous()

    def with_padded_set_to(self, value_to_set: float) -> "RaggedBatch":
        """Copy whose padding holds ``value_to_set`` (out of place)."""
        out = self.as_self_with_cloned_data()
        out.set_padded_to(value_to_set)
        return out

    def set_padded_to(self, value_to_set: float) -> None:
        """Overwrite the padding of ``tensor`` with ``value_to_set`` (in place when ``tensor`` is contiguous with
        the non-uniform dimension right after the batch dimensions; otherwise ``tensor`` is replaced)."""
        t = self._tensor
        moved = self._non_uniform_dim != self._num_batch_dims
        if moved:
            t = t.transpose(self._num_batch_dims, self._non_uniform_dim)
        t = SetPaddedTo.apply(t, self.sample_sizes, value_to_set)
        if moved:
            t = t.transpose(self._num_batch_dims, self._non_uniform_dim)
        self._tensor = t

    def repeat_samples(self, num_repeats: Union[int, Sequence[int]], batch_dim: Optional[int] = None) -> "RaggedBatch":
        """Tile along the batch dimensions: an int repeats ``batch_dim`` (default 0), a sequence gives one
        factor per batch dimension."""
        nb = self._num_batch_dims
        if isinstance(num_repeats, int):
            if batch_dim is None:
                batch_dim = 0
            assert 0 <= batch_dim < nb, f"batch_dim must be in range [0, {nb})"
            reps = [1] * nb
            reps[batch_dim] = num_repeats
        else:
            reps = [int(r) for r in num_repeats]
            assert len(reps) == nb, f"num_repeats must be a sequence of length {nb}"
            assert batch_dim is None, "batch_dim must be None if num_repeats is a sequence"
        tensor = self._tensor.repeat(reps + [1] * (self._tensor.dim() - nb))
        mask = self._mask.repeat(reps + [1]) if self._mask is not None else None
        sizes = self._sample_sizes.repeat(reps) if self._sample_sizes is not None else None
        return RaggedBatch(tensor, mask, sizes, self._non_uniform_dim)

    def unsqueeze_batch_dim(self, dim: int) -> "RaggedBatch":
        """Insert a batch dimension of size 1 at ``dim`` in [0, num_batch_dims]."""
        assert 0 <= dim <= self._num_batch_dims, f"dim must be in range [0, {self._num_batch_dims}]"
        return RaggedBatch(self._tensor.unsqueeze(dim),
                           self._mask.unsqueeze(dim) if self._mask is not None else None,
                           self._sample_sizes.unsqueeze(dim) if self._sample_sizes is not None else None,
                           self._non_uniform_dim + 1)

    def squeeze_batch_dim(self, batch_dim: int) -> "RaggedBatch":
        """Drop batch dimension ``batch_dim`` (must have size 1; at least one batch dimension must remain)."""
        assert 0 <= batch_dim < self._num_batch_dims, f"batch_dim must be in range [0, {self._num_batch_dims})"
        if self._batch_shape[batch_dim] > 1:
            raise ValueError(f"Batch dimension {batch_dim} has size {self._batch_shape[batch_dim]} > 1. Cannot squeeze.")
        return RaggedBatch(self._tensor.squeeze(batch_dim),
                           self._mask.squeeze(batch_dim) if self._mask is not None else None,
                           self._sample_sizes.squeeze(batch_dim) if self._sample_sizes is not None else None,
                           self._non_uniform_dim - 1)

    def reshape_batch_dims(self, new_batch_shape) -> "RaggedBatch":
        """Reshape the batch dimensions (``-1`` allowed); the non-uniform dimension index follows."""
        new_batch_shape = _as_tuple(new_batch_shape)
        nb = self._num_batch_dims
        if new_batch_shape.count(-1) == 1:
            # resolve the free extent from the number of samples: with every sample empty the tensor has no elements and
            # torch cannot infer it (the same reshape in the reference raises there, ragged_batch.py:697)
            known = -math.prod(new_batch_shape)
            total = math.prod(self._batch_shape)
            if known > 0 and total % known == 0:
                new_batch_shape = tuple(total // known if e == -1 else e for e in new_batch_shape)
        tensor = self._tensor.reshape(*new_batch_shape, *self._tensor.shape[nb:])
        mask = self._mask.reshape(*new_batch_shape, self._mask.shape[-1]) if self._mask is not None else None
        sizes = self._sample_sizes.reshape(*new_batch_shape) if self._sample_sizes is not None else None
        return RaggedBatch(tensor, mask, sizes, self._non_uniform_dim - nb + len(new_batch_shape))

    def flatten_batch_dims(self) -> "RaggedBatch":
        return self.reshape_batch_dims(-1)

    def broadcast_batch_dims_to_shape(self, new_batch_shape: Sequence[int]) -> "RaggedBatch":
        """Repeat samples so the batch shape becomes ``new_batch_shape`` (each extent a multiple of the old)."""
        target = _as_tuple(new_batch_shape)
        assert len(target) == self._num_batch_dims, (
            f"New batch shape {target} has {len(target)} dimensions, but {self._num_batch_dims} dimensions are expected.")
        reps = []
        for new, old in zip(target, self._batch_shape):
            assert old > 0 and new % old == 0, f"Cannot broadcast batch dimensions of {self._batch_shape} to {target}."
            reps.append(new // old)
        return self.repeat_samples(reps)

    @staticmethod
    def broadcast_batch_dims(data: Sequence["RaggedBatch"]) -> List["RaggedBatch"]:
        """Bring several instances to their common (element-wise maximum) batch shape."""
        ranks = {d.num_batch_dims for d in data}
        assert len(ranks) == 1, "Cannot broadcast as number of batch dimensions does not match."
        target = tuple(max(d.batch_shape[i] for d in data) for i in range(ranks.pop()))
        return [d.broadcast_batch_dims_to_shape(target) for d in data]

    # ------------------------------------------------------------------ device / dtype
    def to_device(self, device) -> "RaggedBatch":
        return RaggedBatch(self._tensor.to(device=device),
                           self._mask.to(device=device) if self._mask is not None else None,
                           self._sample_sizes.to(device=device) if self._sample_sizes is not None else None,
                           self._non_uniform_dim)

    def cpu(self) -> "RaggedBatch":
        return self.to_device(torch.device("cpu"))

    def to_dtype(self, dtype: torch.dtype) -> "RaggedBatch":
        return RaggedBatch(self._tensor.to(dtype=dtype), self._mask, self._sample_sizes, self._non_uniform_dim)

    def detach(self) -> "RaggedBatch":
        return RaggedBatch(self._tensor.detach(), self._mask, self._sample_sizes, self._non_uniform_dim)

    def to(self, *args, **kwargs) -> "RaggedBatch":
        """``tensor.to(*args, **kwargs)``; mask and sizes follow to the new device if it changed."""
        return self.create_with_sample_sizes_like_self(self._tensor.to(*args, **kwargs))

    def int(self):
        return self.create_with_sample_sizes_like_self(self._tensor.int())

    def long(self):
        return self.create_with_sample_sizes_like_self(self._tensor.long())

    def bool(self):
        return self.create_with_sample_sizes_like_self(self._tensor.bool())

    def half(self):
        return self.create_with_sample_sizes_like_self(self._tensor.half())

    def bfloat16(self):
        return self.create_with_sample_sizes_like_self(self._tensor.bfloat16())

    def float(self):
        return self.create_with_sample_sizes_like_self(self._tensor.float())

    def double(self):
        return self.create_with_sample_sizes_like_self(self._tensor.double())

    def cfloat(self):
        return self.create_with_sample_sizes_like_self(self._tensor.cfloat())

    def cdouble(self):
        return self.create_with_sample_sizes_like_self(self._tensor.cdouble())

    # ------------------------------------------------------------------ data manipulation
    def apply(self, proc_step: Callable):
        """Run ``proc_step(tensor[, mask[, sample_sizes]])`` (arity decides what is passed) and wrap every
        returned tensor with this batch's sizes.  The callable must keep the batch shape, the non-uniform
        dimension and the valid-first ordering."""
        try:
            nargs = proc_step.__code__.co_argcount
        except AttributeError:
            nargs = len(inspect.signature(proc_step).parameters)
        if nargs == 1:
            out = proc_step(self._tensor)
        elif nargs == 2:
            out = proc_step(self._tensor, self.mask)
        elif nargs == 3:
            out = proc_step(self._tensor, self.mask, self.sample_sizes)
        else:
            raise ValueError(f"Function {proc_step} has {nargs} arguments, but only 1, 2, or 3 are supported.")
        wrap = lambda t: RaggedBatch(t, self._mask, self._sample_sizes, self._non_uniform_dim)  # noqa: E731
        return tuple(wrap(t) for t in out) if isinstance(out, tuple) else wrap(out)

    def set_tensor(self, tensor: torch.Tensor) -> None:
        nb = self._num_batch_dims
        assert tensor.shape[:nb] == self._tensor.shape[:nb], (
            f"Batch shape of data to set {tensor.shape[:nb]} does not match current batch shape {self._tensor.shape[:nb]}.")
        assert tensor.shape[self._non_uniform_dim] == self.max_sample_size, (
            f"Maximum sample size of data to set ({tensor.shape[self._non_uniform_dim]}) does not match current "
            f"maximum sample size ({self.max_sample_size}).")
        assert tensor.device == self._tensor.device, (
            f"Device of the data to set ({tensor.device}) does not match current device ({self._tensor.device}).")
        self._tensor = tensor

    def split(self):
        """Un-pad: nested python lists mirroring the batch dimensions, each leaf the sample's valid entries
        (a view of ``tensor``).  Sizes are read back from the device ONCE (the reference indexes the size
        tensor per sample, ragged_batch.py:916-919)."""
        nb = self._num_batch_dims
        src = self if self._non_uniform_dim == nb else self.get_non_uniform_dimension_transposed_to(nb)
        data = src.tensor
        sizes = host_sizes(src.sample_sizes)     # no device read-back when the sizes came from the host (combine_data)
        back = self._non_uniform_dim - nb
        # (explicit batch extent: reshape(-1, ...) is ambiguous when every sample is empty and the tensor has no elements)
        flat = data.reshape(math.prod(data.shape[:nb]), *data.shape[nb:]) if nb > 1 else data
        width = flat.shape[1] if flat.dim() > 1 else 0

        if _bh is not None and flat.dim() > 1:
            leaves = _bh.split_views(flat, sizes, back)   # the per-sample view loop in C++
        else:
            leaves = []
            for s, n in zip(flat.unbind(0), sizes):
                if n != width:
                    s = s.narrow(0, 0, n)
                leaves.append(s.transpose(0, back) if back else s)

        def nest(items, shape):
            if len(shape) == 1:
                return items
            step = len(items) // shape[0] if shape[0] else 0
            return [nest(items[k * step:(k + 1) * step], shape[1:]) for k in range(shape[0])]

        return nest(leaves, tuple(self._batch_shape))

    def unsqueeze_data_dim(self, dim: int) -> "RaggedBatch":
        """Insert a data dimension of size 1 at ``dim`` (>= num_batch_dims; negative counts from the end)."""
        if dim < 0:
            dim += self._tensor.dim() + 1
            assert 0 <= dim <= self._tensor.dim(), "Dimension outside the available range"
        assert dim >= self._num_batch_dims, "Can only add dimensions after the batch dimensions"
        nu = self._non_uniform_dim + (1 if dim <= self._non_uniform_dim else 0)
        return self.create_with_sample_sizes_like_self(self._tensor.unsqueeze(dim), nu)

    # ------------------------------------------------------------------ tensor-like conveniences
    def __getitem__(self, item):
        return self._tensor[item]

    def __setitem__(self, item, value) -> None:
        self._tensor[item] = value

    @property
    def device(self) -> torch.device:
        return self._tensor.device

    @property
    def shape(self) -> torch.Size:
        return self._tensor.shape

    @property
    def dtype(self) -> torch.dtype:
        return self._tensor.dtype

    @property
    def requires_grad(self) -> bool:
        return self._tensor.requires_grad

    @requires_grad.setter
    def requires_grad(self, value: bool) -> None:
        self._tensor.requires_grad = value

    def retain_grad(self) -> None:
        self._tensor.retain_grad()

    @property
    def retains_grad(self) -> bool:
        return self._tensor.retains_grad

    def size(self, *args, **kwargs):
        return self._tensor.size(*args, **kwargs)

    def dim(self) -> int:
        return self._tensor.dim()

    def __repr__(self) -> str:
        m = "*uninitialized*" if self._mask is None else repr(self._mask)
        s = "*uninitialized*" if self._sample_sizes is None else repr(self._sample_sizes)
        return (f"RaggedBatch(tensor={self._tensor}, mask={m}, sample_sizes={s}, "
                f"non_uniform_dim={self._non_uniform_dim}, batch_shape={tuple(self._batch_shape)})")
