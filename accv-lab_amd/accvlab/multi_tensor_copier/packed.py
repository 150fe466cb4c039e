"""DataLoader hook of the multi-tensor copier: pack in the WORKER process, ship ONE buffer.

A batch of thousands of small CPU tensors is expensive long before the host->GPU copy: ``torch.utils.data.DataLoader``
moves every tensor of a worker's batch through its own shared-memory segment (one file descriptor each), and
``pin_memory=True`` pins them one by one.  ``pack_batch`` (or ``packing_collate`` as ``collate_fn``) does the copier's
packing step inside the worker instead: all small contiguous tensors of the batch are laid out with the copier's
pack planner (``accv_mtc_plan``, the byte layout of the reference's compute_pack_plan, multi_tensor_copier.cpp:
419-590) in ONE ``uint8`` buffer; the nesting travels as a flat op list and a few numpy arrays.  The consumer
process receives one tensor, ``pin_memory`` pins one buffer, and ``start_copy(packed_batch, device)`` issues a single
host->GPU transfer (directly from the buffer when it is pinned) and rebuilds the structure as typed views of one GPU
storage — the same result ``start_copy`` gives for the unpacked structure.

Extension: the reference packs in the consumer process only (fill_cpu_staging_buffers, multi_tensor_copier.cpp:647-679).
Everything in this module is host code; the worker side makes no HIP call (safe after fork).
"""
from __future__ import annotations

import ctypes
from typing import Any, Callable, Optional

import numpy as np
import torch

from .. import _amd_native as _nat
from . import copier as _copier

_HUGE_CHUNK = 1 << 62


def _need_host():
    if _copier._host is None:
        raise RuntimeError("PackedBatch needs the multi_tensor_copier host extension "
                           "(make -C accv-lab_amd/csrc_host, or __graft_entry__.build())")
    return _copier._host


class PackedBatch:
    """One nested batch with its small tensors packed into a single byte buffer.  Create with :func:`pack_batch`.

    ``buffer``: ``uint8`` CPU tensor holding the packed leaves; ``num_tensors`` / ``num_packed``: tensor leaves in
    total / inside the buffer.  ``unpack()`` returns the original structure (packed leaves are views of ``buffer``);
    ``pin_memory()`` is what ``DataLoader(pin_memory=True)`` calls on custom batch types.
    """

    __slots__ = ("buffer", "alignment", "_kinds", "_args", "_objects", "_num_leaves", "_leaf_ids", "_offsets", "_dtypes",
                 "_ndims", "_shapes", "_other_ids", "_others")

    def __init__(self, buffer, alignment, kinds, args, objects, num_leaves, leaf_ids, offsets, dtypes, ndims, shapes,
                 other_ids, others):
        self.buffer = buffer
        self.alignment = int(alignment)
        self._kinds, self._args, self._objects, self._num_leaves = kinds, args, objects, int(num_leaves)
        self._leaf_ids, self._offsets, self._dtypes, self._ndims, self._shapes = leaf_ids, offsets, dtypes, ndims, shapes
        self._other_ids, self._others = other_ids, others

    # -- pickling (DataLoader worker -> consumer): slots, tensors go through torch's shared-memory reducers
    def __getstate__(self):
        return tuple(getattr(self, s) for s in self.__slots__)

    def __setstate__(self, state):
        for s, v in zip(self.__slots__, state):
            setattr(self, s, v)

    @property
    def num_tensors(self) -> int:
        return self._num_leaves

    @property
    def num_packed(self) -> int:
        return int(len(self._leaf_ids))

    def is_pinned(self) -> bool:
        return self.buffer.is_pinned()

    def pin_memory(self, device=None) -> "PackedBatch":
        others = [t.pin_memory() if not t.is_cuda else t for t in self._others]
        buf = self.buffer.pin_memory() if self.buffer.numel() else self.buffer
        return PackedBatch(buf, self.alignment, self._kinds, self._args, self._objects, self._num_leaves, self._leaf_ids,
                           self._offsets, self._dtypes, self._ndims, self._shapes, self._other_ids, others)

    def _tree(self, cpu_views: bool):
        """_mtc_host.Tree of this batch; packed leaves are views of ``buffer`` (cpu_views) or left to the caller."""
        tree = _need_host().Tree.from_spec(self._kinds, self._args, self._objects, self._num_leaves)
        for i, t in zip(self._other_ids.tolist(), self._others):
            tree.set_leaf(int(i), t)
        if cpu_views and len(self._leaf_ids):
            tree.views_on(self.buffer, 0, self._leaf_ids, self._offsets, self._dtypes, self._ndims, self._shapes, True)
        return tree

    def unpack(self):
        """The original nested structure on the CPU; packed tensors alias ``buffer`` (zero copy)."""
        tree = self._tree(cpu_views=True)
        for i, t in zip(self._other_ids.tolist(), self._others):
            tree.set_out(int(i), t)
        return _copier._rebuild_without_gc(tree)


def pack_batch(data: Any, *, min_packed_alignment_bytes: int = 16) -> PackedBatch:
    """Pack the small contiguous CPU tensors (0 < bytes <= 256 KiB, any dtype mix) and numpy arrays of a nested
    list/tuple/dict structure into one buffer.  Other tensors (large, non-contiguous, already on a GPU) ride along
    unpacked; non-tensor leaves are kept by reference.  Meant to run in DataLoader workers (see :func:`packing_collate`).
    """
    host = _need_host()
    lib = _nat.lib()
    tree = host.Tree(data)
    n = tree.num_leaves()
    route, nbytes, esize, ptr, _ = tree.classify("cuda:0", True)   # "what would a host->GPU copy pack?"; no GPU touched
    packable = np.nonzero(route == _copier.R_H2D_PACK)[0].astype(np.int64)
    min_align = max(1, int(min_packed_alignment_bytes))
    align = 16
    while align < min_align:
        align <<= 1
    total, off = 0, np.empty(0, dtype=np.int64)
    if len(packable) >= 2:
        m = len(packable)
        nb = np.ascontiguousarray(nbytes[packable], dtype=np.int64)
        es = np.ascontiguousarray(esize[packable], dtype=np.int32)
        cand = np.ones(m, dtype=np.uint8)
        off = np.empty(m, dtype=np.int64)
        chk = np.empty(m, dtype=np.int64)
        csz = np.empty(m, dtype=np.int64)
        nck = ctypes.c_longlong(0)
        _nat.check(lib.accv_mtc_plan(m, nb.ctypes.data, es.ctypes.data, cand.ctypes.data, min_align, _HUGE_CHUNK,
                                     off.ctypes.data, chk.ctypes.data, csz.ctypes.data, ctypes.addressof(nck)),
                   "mtc_plan")
        assert nck.value == 1
        total = int(csz[0])
    else:
        packable = packable[:0]
    buffer = torch.empty(total, dtype=torch.uint8)
    if total:
        src = np.ascontiguousarray(ptr[packable], dtype=np.uint64)
        _nat.check(lib.accv_mtc_pack_host(len(packable), src.ctypes.data, nb.ctypes.data, off.ctypes.data,
                                          buffer.data_ptr(), total), "mtc_pack_host")
    dtypes, ndims, shapes = tree.leaf_meta(packable)
    is_other = np.ones(n, dtype=bool)
    is_other[packable] = False
    other_ids = np.nonzero(is_other)[0].astype(np.int64)
    others = [tree.leaf(int(i)) for i in other_ids.tolist()]
    kinds, args, objects = tree.export_spec()
    return PackedBatch(buffer, align, kinds, args, objects, n, packable, off, dtypes, ndims, shapes, other_ids, others)


def packing_collate(collate_fn: Optional[Callable] = None, *, min_packed_alignment_bytes: int = 16) -> Callable:
    """``collate_fn`` for ``torch.utils.data.DataLoader``: runs ``collate_fn`` (default: keep the list of samples) in
    the worker and returns its result as a :class:`PackedBatch`::

        loader = DataLoader(ds, batch_size=8, num_workers=4, pin_memory=True, collate_fn=packing_collate())
        for packed in loader:
            batch = start_copy(packed, "cuda:0").get()
    """
    return _PackingCollate(collate_fn, min_packed_alignment_bytes)


class _PackingCollate:  # a picklable callable (spawn start method)
    def __init__(self, inner, min_align):
        self.inner = inner
        self.min_align = min_align

    def __call__(self, samples):
        return pack_batch(self.inner(samples) if self.inner is not None else samples,
                          min_packed_alignment_bytes=self.min_align)
