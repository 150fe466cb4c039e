"""start_copy / AsyncCopyHandle — nested-structure tensor copier with small-tensor coalescing.

Same public contract as the reference (packages/multi_tensor_copier/accvlab/multi_tensor_copier/async_copy.py:27-169
on top of csrc/multi_tensor_copier.cpp:863-883, 922-1065, 1074-1158):

* list / tuple / dict containers are walked; torch tensors and numpy arrays are copied to ``device``; every other
  leaf is returned by identity; container types are preserved;
* host->GPU: contiguous host tensors of 0 < bytes <= 256 KiB are packed (mixed dtypes) into aligned byte chunks of at
  most ``max_packed_chunk_bytes``; each chunk is staged in pinned memory by a native worker pool and moved with ONE
  hipMemcpyAsync; the results are typed views that share one GPU storage per chunk;
* GPU->host with ``use_pinned_staging`` returns pinned tensors; tensors already on the target device are returned
  as they are; GPU->GPU and host->host are supported;
* copies are ordered after the work that was enqueued on the caller's current stream at call time; ``get()`` blocks
  until the data is usable from any stream; worker exceptions surface from ``ready()`` / ``get()``.

MI355X-first differences: pinned staging comes from a recycled arena (accv_pinned_*) instead of a fresh pinned
allocation per call, and the transfers run on a dedicated non-blocking side stream that waits on an event of the
caller's stream (the reference enqueues on the caller's stream itself) so the DMA overlaps the caller's kernels.
"""
from __future__ import annotations

import ctypes
import os
import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Any, List, Optional, Tuple

import numpy as np
import torch

from .. import _amd_native as _nat

PACK_MAX_BYTES_PER_TENSOR = 256 * 1024  # multi_tensor_copier.cpp:483

_pool_lock = threading.Lock()
_pool: Optional[ThreadPoolExecutor] = None
_side_streams = {}


def _executor() -> ThreadPoolExecutor:
    """Process-wide pool of <= 4 orchestration threads (reference: CopyThreadPool, multi_tensor_copier.cpp:288-349)."""
    global _pool
    with _pool_lock:
        if _pool is None:
            _pool = ThreadPoolExecutor(max_workers=max(1, min(4, os.cpu_count() or 1)), thread_name_prefix="accv-mtc")
        return _pool


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    with _pool_lock:
        s = _side_streams.get(device.index)
        if s is None:
            s = torch.cuda.Stream(device=device)
            _side_streams[device.index] = s
        return s


# ------------------------------------------------------------------------------------------------ tree walk
_T, _L, _D, _X, _P = 0, 1, 2, 3, 4  # tuple, list, dict, tensor leaf, passthrough


def _flatten(obj: Any, leaves: List[torch.Tensor]):
    """Returns a spec; appends tensor leaves (numpy arrays become CPU tensors that share memory when possible)."""
    if isinstance(obj, torch.Tensor):
        leaves.append(obj)
        return (_X, len(leaves) - 1)
    t = type(obj)
    if t is list:
        return (_L, [_flatten(o, leaves) for o in obj])
    if t is tuple:
        return (_T, [_flatten(o, leaves) for o in obj])
    if t is dict:
        return (_D, list(obj.keys()), [_flatten(o, leaves) for o in obj.values()])
    if isinstance(obj, np.ndarray):
        try:
            ten = torch.from_numpy(obj)
        except (TypeError, ValueError, RuntimeError):
            ten = torch.from_numpy(np.ascontiguousarray(obj).copy())
        leaves.append(ten)
        return (_X, len(leaves) - 1)
    return (_P, obj)


def _rebuild(spec, outs: List[torch.Tensor]):
    kind = spec[0]
    if kind == _X:
        return outs[spec[1]]
    if kind == _P:
        return spec[1]
    if kind == _L:
        return [_rebuild(s, outs) for s in spec[1]]
    if kind == _T:
        return tuple(_rebuild(s, outs) for s in spec[1])
    return {k: _rebuild(s, outs) for k, s in zip(spec[1], spec[2])}


# ------------------------------------------------------------------------------------------------ job
class _Job:
    """Everything one start_copy call owns until the handle is consumed."""

    def __init__(self, spec, leaves, device, pinned, pack, min_align, max_chunk):
        self.spec = spec
        self.leaves = leaves            # keeps the inputs alive
        self.device = device
        self.pinned = pinned
        self.pack = pack
        self.min_align = max(1, int(min_align))
        self.max_chunk = int(max_chunk)
        self.outs: List[Optional[torch.Tensor]] = [None] * len(leaves)
        self.events: List[torch.cuda.Event] = []
        self.staging: List[int] = []    # arena pointers to give back once the DMA is done
        self.keep: List[Any] = []       # chunk tensors / pinned intermediates
        self.caller_stream = None
        self.source_events = {}
        self.released = False

    def release_staging(self):
        if not self.released:
            self.released = True
            lib = _nat.lib()
            for p in self.staging:
                lib.accv_pinned_release(p)
            self.staging = []


def _contig_strides(t: torch.Tensor):
    return t.stride()


def _run(job: _Job) -> None:
    """Orchestration (worker thread or inline): plan, allocate, stage, enqueue, build views."""
    lib = _nat.lib()
    dev = job.device
    leaves = job.leaves
    n = len(leaves)
    to_gpu = dev.type == "cuda"

    h2d_packable, h2d_single, d2h, d2d = [], [], [], []
    for i, t in enumerate(leaves):
        if t.device == dev:
            job.outs[i] = t                      # reuse as is (multi_tensor_copier.cpp:783-786)
        elif t.device.type == "cpu" and to_gpu:
            nbytes = t.numel() * t.element_size()
            if job.pack and t.is_contiguous() and 0 < nbytes <= PACK_MAX_BYTES_PER_TENSOR:
                h2d_packable.append(i)
            else:
                h2d_single.append(i)
        elif t.device.type == "cuda" and dev.type == "cpu":
            d2h.append(i)
        elif t.device.type == "cuda" and to_gpu:
            d2d.append(i)
        else:
            job.outs[i] = t.to(dev)

    if to_gpu:
        side = _side_stream(dev)
        with torch.cuda.device(dev):
            # ---- packed host -> GPU
            n_chunks = 0
            if len(h2d_packable) >= 2:
                m = len(h2d_packable)
                nbytes = np.fromiter((leaves[i].numel() * leaves[i].element_size() for i in h2d_packable), np.int64, m)
                esize = np.fromiter((leaves[i].element_size() for i in h2d_packable), np.int32, m)
                cand = np.ones(m, dtype=np.uint8)
                off = np.empty(m, dtype=np.int64)
                chk = np.empty(m, dtype=np.int64)
                csz = np.empty(m, dtype=np.int64)
                nck = ctypes.c_longlong(0)
                _nat.check(lib.accv_mtc_plan(m, nbytes.ctypes.data, esize.ctypes.data, cand.ctypes.data, job.min_align,
                                             job.max_chunk, off.ctypes.data, chk.ctypes.data, csz.ctypes.data,
                                             ctypes.addressof(nck)), "mtc_plan")
                n_chunks = int(nck.value)
            if n_chunks > 0:
                align = 16
                while align < job.min_align:
                    align <<= 1                    # packed_buffer_alignment_bytes, multi_tensor_copier.cpp:399-404
                order = np.argsort(chk, kind="stable").astype(np.int64)
                begin = np.searchsorted(chk[order], np.arange(n_chunks + 1)).astype(np.int64)
                src = np.fromiter((leaves[i].data_ptr() for i in h2d_packable), np.uint64, m)
                stage_ptrs = np.empty(n_chunks, dtype=np.uint64)
                dev_ptrs = np.empty(n_chunks, dtype=np.uint64)
                chunks = []
                with torch.cuda.stream(job.caller_stream):
                    for c in range(n_chunks):
                        size = int(csz[c])
                        g = torch.empty(size + align + 15, dtype=torch.uint8, device=dev)
                        base = (-g.data_ptr()) % align
                        chunks.append((g, base))
                        dev_ptrs[c] = g.data_ptr() + base
                        if job.pinned:
                            p = lib.accv_pinned_acquire(size)
                            if not p:
                                _nat.check(-4, "pinned arena")
                            job.staging.append(p)
                            stage_ptrs[c] = p
                        else:
                            buf = torch.empty(size, dtype=torch.uint8)
                            job.keep.append(buf)
                            stage_ptrs[c] = buf.data_ptr()
                    ready = torch.cuda.Event()
                    ready.record(job.caller_stream)
                side.wait_event(ready)            # after the caller's work AND after the allocation point
                _nat.check(lib.accv_mtc_stage_h2d(m, src.ctypes.data, nbytes.ctypes.data, off.ctypes.data,
                                                  order.ctypes.data, n_chunks, begin.ctypes.data, stage_ptrs.ctypes.data,
                                                  dev_ptrs.ctypes.data, csz.ctypes.data, side.cuda_stream, 0),
                           "mtc_stage_h2d")
                # typed views into the chunk storage (enqueue_packed_transfer, multi_tensor_copier.cpp:712-729)
                typed = {}
                for k, i in enumerate(h2d_packable):
                    t = leaves[i]
                    c = int(chk[k])
                    g, base = chunks[c]
                    key = (c, t.dtype)
                    tv = typed.get(key)
                    if tv is None:
                        es = t.element_size()
                        usable = (g.numel() - base) // 16 * 16
                        tv = g[base:base + usable].view(t.dtype)
                        typed[key] = tv
                    job.outs[i] = torch.as_strided(tv, t.shape, t.stride(), int(off[k]) // t.element_size())
                job.keep.append(chunks)
            else:
                h2d_single = sorted(h2d_single + h2d_packable)
            # ---- everything else that targets the GPU goes through torch on the side stream
            if h2d_single or d2d:
                ready = torch.cuda.Event()
                ready.record(job.caller_stream)
                side.wait_event(ready)
                for dev_idx, ev in job.source_events.items():
                    side.wait_event(ev)           # synchronize_source_streams, multi_tensor_copier.cpp:741-762
                with torch.cuda.stream(job.caller_stream):
                    for i in h2d_single + d2d:
                        job.outs[i] = torch.empty_like(leaves[i], device=dev)
                with torch.cuda.stream(side):
                    for i in h2d_single:
                        src_t = leaves[i]
                        if job.pinned and not src_t.is_pinned():
                            src_t = src_t.contiguous().pin_memory()
                            job.keep.append(src_t)
                        job.outs[i].copy_(src_t, non_blocking=True)
                    for i in d2d:
                        job.outs[i].copy_(leaves[i], non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
            job.events.append(done)
    if d2h:
        by_dev = {}
        for i in d2h:
            by_dev.setdefault(leaves[i].device, []).append(i)
        for sdev, idxs in by_dev.items():
            side = _side_stream(sdev)
            with torch.cuda.device(sdev):
                side.wait_event(job.source_events[sdev.index])
                small = [i for i in idxs if job.pack and leaves[i].is_contiguous()
                         and 0 < leaves[i].numel() * leaves[i].element_size() <= PACK_MAX_BYTES_PER_TENSOR]
                packed_ok = False
                if len(small) >= 2:
                    packed_ok = _coalesced_d2h(job, lib, small, sdev, side)
                rest = [i for i in idxs if not (packed_ok and i in set(small))]
                with torch.cuda.stream(side):
                    for i in rest:
                        t = leaves[i]
                        out = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=job.pinned)
                        out.copy_(t, non_blocking=job.pinned)
                        job.outs[i] = out
                done = torch.cuda.Event()
                done.record(side)
                job.events.append(done)


def _coalesced_d2h(job: _Job, lib, small: List[int], sdev: torch.device, side) -> bool:
    """Many small device tensors -> ONE device gather kernel (accv_mtc_coalesce) -> ONE D2H transfer per chunk -> host
    views (the reference copies each tensor separately, multi_tensor_copier.cpp:790-800)."""
    leaves = job.leaves
    m = len(small)
    nbytes = np.fromiter((leaves[i].numel() * leaves[i].element_size() for i in small), np.int64, m)
    esize = np.fromiter((leaves[i].element_size() for i in small), np.int32, m)
    off = np.empty(m, dtype=np.int64)
    chk = np.empty(m, dtype=np.int64)
    csz = np.empty(m, dtype=np.int64)
    nck = ctypes.c_longlong(0)
    _nat.check(lib.accv_mtc_plan(m, nbytes.ctypes.data, esize.ctypes.data, np.ones(m, dtype=np.uint8).ctypes.data,
                                 job.min_align, job.max_chunk, off.ctypes.data, chk.ctypes.data, csz.ctypes.data,
                                 ctypes.addressof(nck)), "mtc_plan")
    n_chunks = int(nck.value)
    if n_chunks == 0:
        return False
    ptrs = np.fromiter((leaves[i].data_ptr() for i in small), np.uint64, m)
    with torch.cuda.stream(side):
        for c in range(n_chunks):
            sel = np.nonzero(chk == c)[0]
            size = int(csz[c])
            table = np.empty((len(sel), 3), dtype=np.int64)      # {src pointer, offset in packed, nbytes}
            table[:, 0] = ptrs[sel].view(np.int64)
            table[:, 1] = off[sel]
            table[:, 2] = nbytes[sel]
            items = torch.from_numpy(table).to(sdev, non_blocking=False)
            packed = torch.empty(size + 16, dtype=torch.uint8, device=sdev)
            _nat.check(lib.accv_mtc_coalesce(items.data_ptr(), len(sel), packed.data_ptr(), 0, side.cuda_stream),
                       "mtc_coalesce")
            host = torch.empty(size + 16, dtype=torch.uint8, device="cpu", pin_memory=job.pinned)
            host.copy_(packed, non_blocking=job.pinned)
            job.keep.append((items, packed))
            typed = {}
            for k in sel:
                t = leaves[small[k]]
                tv = typed.get(t.dtype)
                if tv is None:
                    tv = host[: (size + 15) // 16 * 16].view(t.dtype)
                    typed[t.dtype] = tv
                job.outs[small[k]] = torch.as_strided(tv, t.shape, t.stride(), int(off[k]) // t.element_size())
    return True


class AsyncCopyHandle:
    """Handle to an in-progress copy started by :func:`start_copy`.  ``ready()`` polls, ``get()`` blocks and returns
    the copied structure.  Dropping the handle early blocks in the destructor until the transfers are done so that
    staging buffers are never recycled under an active DMA (reference: multi_tensor_copier.cpp:932-943)."""

    def __init__(self, job: _Job, future: Optional[Future]):
        self._job = job
        self._future = future
        self._result = None
        self._consumed = False

    def _finish(self) -> None:
        if self._future is not None:
            self._future.result()            # re-raises worker exceptions
        for ev in self._job.events:
            ev.synchronize()
        self._job.release_staging()

    def ready(self) -> bool:
        """True once the copy has completed (non-blocking).  Raises if the copy failed."""
        if self._future is not None:
            if not self._future.done():
                return False
            self._future.result()
        if all(ev.query() for ev in self._job.events):
            self._job.release_staging()
            return True
        return False

    def get(self):
        """Block until done; return the input structure with every tensor on the target device."""
        if not self._consumed:
            self._finish()
            self._result = _rebuild(self._job.spec, self._job.outs)
            self._consumed = True
        return self._result

    def __del__(self):
        try:
            if not self._consumed:
                self._finish()
        except Exception:
            pass


def start_copy(data, device, *, use_pinned_staging: bool = True, pack_cpu_tensors: bool = True,
               min_packed_alignment_bytes: int = 16, max_packed_chunk_bytes: int = 32 * 1024 * 1024,
               use_background_thread: bool = True) -> AsyncCopyHandle:
    """Asynchronously copy all tensors of a nested list/tuple/dict structure to ``device``.

    Args:
        data: a tensor / numpy array or a nesting of ``list`` / ``tuple`` / ``dict`` with tensor, array and arbitrary
            other leaves (the latter are passed through unchanged; other container types count as opaque leaves).
        device: target device (``"cuda:0"``, ``"cpu"``, ``torch.device``).
        use_pinned_staging: stage host<->GPU transfers through pinned memory (host->GPU becomes asynchronous;
            GPU->host returns the pinned buffer itself).
        pack_cpu_tensors: coalesce small contiguous host tensors (<= 256 KiB each, any dtype mix) into shared chunks
            with one transfer per chunk.  Only for host->GPU.
        min_packed_alignment_bytes: each packed tensor starts at a multiple of
            ``round_up(max(min_packed_alignment_bytes, element_size), element_size)`` inside its chunk.
        max_packed_chunk_bytes: payload limit per chunk (default 32 MiB).
        use_background_thread: run allocation, staging and enqueueing on a pool thread so this call returns early.

    The inputs must stay alive and unmodified until ``get()`` returned or ``ready()`` was True.
    """
    dev = torch.device(device)  # RuntimeError on malformed strings (reference :225-232)
    if dev.type == "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError(f"Invalid device string: {device!r} (no GPU is available)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        _nat.lib()  # fail loudly if the HIP library is missing
    elif dev.type != "cpu":
        raise RuntimeError(f"Invalid device string: {device!r}")
    leaves: List[torch.Tensor] = []
    spec = _flatten(data, leaves)
    job = _Job(spec, leaves, dev, bool(use_pinned_staging), bool(pack_cpu_tensors), int(min_packed_alignment_bytes),
               int(max_packed_chunk_bytes))
    # ordering: capture the caller's current streams NOW (reference :1086-1123)
    if dev.type == "cuda":
        job.caller_stream = torch.cuda.current_stream(dev)
    for t in leaves:
        if t.device.type == "cuda" and t.device != dev and t.device.index not in job.source_events:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(t.device))
            job.source_events[t.device.index] = ev
    if use_background_thread:
        return AsyncCopyHandle(job, _executor().submit(_run, job))
    _run(job)  # inline: exceptions propagate from start_copy (reference :1151-1153)
    return AsyncCopyHandle(job, None)
