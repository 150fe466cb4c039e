"""start_copy / AsyncCopyHandle — nested-structure tensor copier with small-tensor coalescing.

Same public contract as the reference (packages/multi_tensor_copier/accvlab/multi_tensor_copier/async_copy.py:27-169
on top of csrc/multi_tensor_copier.cpp:863-883, 922-1065, 1074-1158):

* list / tuple / dict containers are walked; torch tensors and numpy arrays are copied to ``device``; every other
  leaf is returned by identity; container types are preserved;
* host->GPU: contiguous host tensors of 0 < bytes <= 256 KiB are packed (mixed dtypes) into aligned byte chunks of at
  most ``max_packed_chunk_bytes``; each chunk is staged in pinned memory by a native worker pool and moved with ONE
  hipMemcpyAsync; the results are typed views that share one GPU storage per chunk;
* GPU->host with ``use_pinned_staging`` returns pinned tensors (many small device tensors travel as ONE device-side
  gather + ONE transfer); tensors already on the target device are returned as they are; GPU->GPU and host->host work;
* copies are ordered after the work that was enqueued on the caller's current stream at call time; ``get()`` blocks
  until the data is usable from any stream; worker exceptions surface from ``ready()`` / ``get()``.

MI355X-first differences: pinned staging comes from a recycled arena (accv_pinned_*) instead of a fresh pinned
allocation per call, and the transfers run on a dedicated non-blocking side stream that waits on an event of the
caller's stream (the reference enqueues on the caller's stream itself) so the DMA overlaps the caller's kernels.

Host overhead: the tree walk, leaf classification, packed-view construction and output rebuild run in the C++ module
``_mtc_host`` (csrc_host/mtc_host.cpp, pybind11 + ATen) when it has been built; ``_PyLeafSet`` is its pure-python twin
(same interface) used otherwise.  Both drive the same C-ABI calls of libaccv_hip.so.
"""
from __future__ import annotations

import ctypes
import gc
import os
import threading
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Any, List, Optional

import numpy as np
import torch

from .. import _amd_native as _nat

try:  # C++ host fast path (built by `make -C accv-lab_amd/csrc_host`)
    from . import _mtc_host as _host
except ImportError:  # pragma: no cover - exercised when the host extension has not been built
    _host = None

PACK_MAX_BYTES_PER_TENSOR = 256 * 1024  # multi_tensor_copier.cpp:483
_NATIVE_THREAD_MIN_BYTES = 2 << 20      # background jobs below this stage on the caller's thread (thread wake-ups cost more)
R_REUSE, R_H2D_PACK, R_H2D_SINGLE, R_D2H_SMALL, R_D2H_OTHER, R_D2D, R_OTHER, R_D2D_SMALL = range(8)

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


# ------------------------------------------------------------------------------------------------ python leaf set
_T, _L, _D, _X, _P = 0, 1, 2, 3, 4  # tuple, list, dict, tensor leaf, passthrough


class _PyLeafSet:
    """Pure-python twin of _mtc_host.Tree (same methods)."""

    def __init__(self, data: Any):
        self._leaves: List[torch.Tensor] = []
        self._spec = self._walk(data)
        self._outs: List[Optional[torch.Tensor]] = [None] * len(self._leaves)

    def _walk(self, obj):
        if isinstance(obj, torch.Tensor):
            self._leaves.append(obj)
            return (_X, len(self._leaves) - 1)
        t = type(obj)
        if t is list:
            return (_L, [self._walk(o) for o in obj])
        if t is tuple:
            return (_T, [self._walk(o) for o in obj])
        if t is dict:
            return (_D, list(obj.keys()), [self._walk(o) for o in obj.values()])
        if isinstance(obj, np.ndarray):
            try:
                ten = torch.from_numpy(obj)
            except (TypeError, ValueError, RuntimeError):
                ten = torch.from_numpy(np.array(obj, order="C"))
            self._leaves.append(ten)
            return (_X, len(self._leaves) - 1)
        return (_P, obj)

    def num_leaves(self) -> int:
        return len(self._leaves)

    def leaf(self, i: int) -> torch.Tensor:
        return self._leaves[i]

    def set_out(self, i: int, t: torch.Tensor) -> None:
        self._outs[i] = t

    def classify(self, device: str, pack: bool):
        dev = torch.device(device)
        n = len(self._leaves)
        route = np.empty(n, dtype=np.int8)
        nbytes = np.empty(n, dtype=np.int64)
        esize = np.empty(n, dtype=np.int32)
        ptr = np.empty(n, dtype=np.uint64)
        didx = np.empty(n, dtype=np.int32)
        for i, t in enumerate(self._leaves):
            b = t.numel() * t.element_size()
            nbytes[i], esize[i] = b, t.element_size()
            ptr[i] = t.data_ptr() if t.numel() else 0
            didx[i] = -1 if t.device.index is None else t.device.index
            small = t.is_contiguous() and 0 < b <= PACK_MAX_BYTES_PER_TENSOR
            if t.device == dev:
                route[i] = R_REUSE
                self._outs[i] = t
            elif t.device.type == "cpu" and dev.type == "cuda":
                route[i] = R_H2D_PACK if (pack and small) else R_H2D_SINGLE
            elif t.device.type == "cuda" and dev.type == "cpu":
                route[i] = R_D2H_SMALL if (pack and small) else R_D2H_OTHER
            elif t.device.type == "cuda" and dev.type == "cuda":
                route[i] = R_D2D_SMALL if (pack and small) else R_D2D
            else:
                route[i] = R_OTHER
        return route, nbytes, esize, ptr, didx

    def make_packed_views(self, idx, chunk_of, offsets, chunks, bases) -> None:
        typed = {}
        for k in range(len(idx)):
            t = self._leaves[int(idx[k])]
            c = int(chunk_of[k])
            g = chunks[c]
            key = (c, t.dtype)
            tv = typed.get(key)
            if tv is None:
                base = int(bases[c])
                usable = (g.numel() - base) // 16 * 16
                tv = g[base:base + usable].view(t.dtype)
                typed[key] = tv
            self._outs[int(idx[k])] = torch.as_strided(tv, t.shape, t.stride(), int(offsets[k]) // t.element_size())

    def rebuild(self):
        def build(spec):
            kind = spec[0]
            if kind == _X:
                out = self._outs[spec[1]]
                if out is None:
                    raise RuntimeError(f"output {spec[1]} was never produced")
                return out
            if kind == _P:
                return spec[1]
            if kind == _L:
                return [build(s) for s in spec[1]]
            if kind == _T:
                return tuple(build(s) for s in spec[1])
            return {k: build(s) for k, s in zip(spec[1], spec[2])}

        return build(self._spec)


def _make_leaf_set(data):
    if _host is not None and os.environ.get("ACCV_MTC_PY_HOST", "0") != "1":
        return _host.Tree(data)
    return _PyLeafSet(data)


# ------------------------------------------------------------------------------------------------ job
class _Job:
    """Everything one start_copy call owns until the handle is consumed."""

    def __init__(self, tree, device, pinned, pack, min_align, max_chunk):
        self.tree = tree                 # keeps the inputs alive
        self.device = device
        self.pinned = pinned
        self.pack = pack
        self.min_align = max(1, int(min_align))
        self.max_chunk = int(max_chunk)
        self.meta = None                 # (route, nbytes, esize, ptr, device index) arrays
        self.events: List[torch.cuda.Event] = []
        self.staging: List[int] = []     # arena pointers to give back once the DMA is done
        self.keep: List[Any] = []        # chunk tensors / pinned intermediates
        self.caller_stream = None
        self.source_events = {}
        self.released = False
        self.packed = None               # PackedBatch whose buffer still has to travel (host->GPU)
        self.ticket = None               # native orchestration job (accv_mtc_stage_h2d_async) still to be waited for
        self.pending_views = None        # arguments of tree.make_packed_views, built by the consumer while the job runs
        self.side = None                 # side stream the native job enqueues on

    def release_staging(self):
        if not self.released:
            self.released = True
            lib = _nat.lib()
            for p in self.staging:
                lib.accv_pinned_release(p)
            self.staging = []


def nbytes_all(job: _Job) -> np.ndarray:
    return job.meta[1]


# the plan of the last call: a training loop copies the same sizes every step, and the plan (offsets, chunk of every leaf,
# chunk sizes, staging order) is a function of the sizes alone — 0.2 of 1.1 ms for 10 000 leaves.  The arrays are never
# written after they were made, so handing the same ones to several jobs is safe.
_last_plan = None


def _plan(lib, job: _Job, nbytes: np.ndarray, esize: np.ndarray):
    global _last_plan
    m = len(nbytes)
    nbytes = np.ascontiguousarray(nbytes, dtype=np.int64)
    esize = np.ascontiguousarray(esize, dtype=np.int32)
    last = _last_plan
    if last is not None and last[0] == (m, job.min_align, job.max_chunk) and np.array_equal(last[1], nbytes) and \
            np.array_equal(last[2], esize):
        return last[3]
    out = _plan_uncached(lib, job, nbytes, esize, m)
    _last_plan = ((m, job.min_align, job.max_chunk), nbytes, esize, out)
    return out


def _plan_uncached(lib, job: _Job, nbytes: np.ndarray, esize: np.ndarray, m: int):
    cand = np.ones(m, dtype=np.uint8)
    off = np.empty(m, dtype=np.int64)
    chk = np.empty(m, dtype=np.int64)
    csz = np.empty(max(m, 1), dtype=np.int64)
    nck = ctypes.c_longlong(0)
    _nat.check(lib.accv_mtc_plan(m, nbytes.ctypes.data, esize.ctypes.data, cand.ctypes.data, job.min_align, job.max_chunk,
                                 off.ctypes.data, chk.ctypes.data, csz.ctypes.data, ctypes.addressof(nck)), "mtc_plan")
    n_chunks = int(nck.value)
    # staging order: the leaves of a chunk one after the other (also a function of the plan alone)
    order = np.argsort(chk, kind="stable").astype(np.int64)
    begin = np.searchsorted(chk[order], np.arange(max(n_chunks, 0) + 1)).astype(np.int64)
    return off, chk, csz, n_chunks, nbytes, order, begin


def _prepare_packed_h2d(job: _Job, lib, packable: np.ndarray, side):
    """Pack plan, GPU chunks (allocated on the caller's stream), pinned staging blocks and the stream ordering for the
    packable host leaves; returns the argument lists of the staging call and of make_packed_views, or None when the
    plan packs nothing (fewer than two candidates)."""
    dev = job.device
    route, nbytes, esize, ptr, didx = job.meta
    off, chk, csz, n_chunks, pbytes, order, begin = _plan(lib, job, nbytes[packable], esize[packable])
    if n_chunks <= 0:
        return None
    m = len(packable)
    align = 16
    while align < job.min_align:
        align <<= 1                    # packed_buffer_alignment_bytes, multi_tensor_copier.cpp:399-404
    src = np.ascontiguousarray(ptr[packable], dtype=np.uint64)
    stage_ptrs = np.empty(n_chunks, dtype=np.uint64)
    dev_ptrs = np.empty(n_chunks, dtype=np.uint64)
    bases = np.empty(n_chunks, dtype=np.int64)
    chunks = []
    with torch.cuda.stream(job.caller_stream):
        for c in range(n_chunks):
            size = int(csz[c])
            g = torch.empty(size + align + 15, dtype=torch.uint8, device=dev)
            bases[c] = (-g.data_ptr()) % align
            chunks.append(g)
            dev_ptrs[c] = g.data_ptr() + int(bases[c])
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
    job.keep.append(chunks)
    return {"stage_args": (m, src.ctypes.data, pbytes.ctypes.data, off.ctypes.data, order.ctypes.data, n_chunks,
                           begin.ctypes.data, stage_ptrs.ctypes.data, dev_ptrs.ctypes.data, csz.ctypes.data),
            "arrays": (src, pbytes, off, order, begin, stage_ptrs, dev_ptrs, csz),     # keep the buffers alive
            "views": (packable, chk, off, chunks, bases)}


def _start_native_h2d(job: _Job) -> bool:
    """Background mode without a Python worker: when every leaf is either reused as it is or a packable host tensor
    (the common case — a batch of small CPU tensors), the caller thread plans and allocates, a NATIVE library thread
    stages and enqueues (accv_mtc_stage_h2d_async: no interpreter lock anywhere in the work), and the consumer builds the
    views while that thread runs.  Returns False when the job needs the general python orchestration."""
    route = job.meta[0]
    if job.device.type != "cuda" or job.packed is not None:
        return False
    packable = np.nonzero(route == R_H2D_PACK)[0].astype(np.int64)
    if len(packable) < 2 or int(np.count_nonzero((route != R_REUSE) & (route != R_H2D_PACK))) > 0:
        return False
    lib = _nat.lib()
    side = _side_stream(job.device)
    with torch.cuda.device(job.device):
        prep = _prepare_packed_h2d(job, lib, packable, side)
        if prep is None:
            return False
        if int(prep["arrays"][1].sum()) <= _NATIVE_THREAD_MIN_BYTES:
            # a few hundred KB stage in tens of microseconds: waking another thread (and being woken by it) costs more
            # than the memcpy — stage here, the transfer itself is asynchronous on the side stream either way
            _nat.check(lib.accv_mtc_stage_h2d(*prep["stage_args"], side.cuda_stream, 0), "mtc_stage_h2d")
            done = torch.cuda.Event()
            done.record(side)
            job.events.append(done)
            job.pending_views = prep["views"]
            return True
        _register_native_shutdown()
        ticket = ctypes.c_longlong(0)
        _nat.check(lib.accv_mtc_stage_h2d_async(*prep["stage_args"], side.cuda_stream, 0, job.device.index,
                                                ctypes.addressof(ticket)), "mtc_stage_h2d_async")
    job.ticket, job.pending_views, job.side = int(ticket.value), prep["views"], side
    return True


_shutdown_registered = False


def _register_native_shutdown() -> None:
    """The library's orchestration thread is stopped from ``atexit`` — i.e. while the HIP runtime is still alive — instead
    of from a static destructor after it (the reference joins its CopyThreadPool in the pool's destructor,
    multi_tensor_copier.cpp:300-312)."""
    global _shutdown_registered
    if not _shutdown_registered:
        import atexit

        atexit.register(lambda: _nat.ctypes_lib().accv_mtc_shutdown())
        _shutdown_registered = True


def set_output_recycling(enabled: bool) -> None:
    """(extension) Switch the re-use of output tensor objects between consecutive packed copies on (default) or off.

    A result of N packed leaves is N tensor objects, and creating / destroying them is what a many-leaf copy costs
    (≈ 300 ns per leaf; the transfer itself is microseconds).  The C++ host path therefore keeps the packed output tensors
    of the last two calls and, when the next call has a packed leaf at the same position with the same dtype and device,
    re-points a kept tensor object at its new place in the new chunk — but ONLY a tensor that nothing refers to any more
    (no variable, container or view in python, nothing in C++).  A result whose leaves are all packed views is kept with its
    containers as well and the very same tree is returned again by a later copy of the same structure (equal keys; non-tensor
    leaves are replaced by the new call's), again only when nothing refers to the tree or any part of it.  Visible effects: such an object may come back with a
    new content under the same ``id()`` (a ``weakref`` to an old output is not a reference), and the chunk storage of
    up to two earlier results stays allocated (≤ 256 MB each) until the next call or :func:`release_cached_outputs`.
    """
    if _host is not None:
        _host.set_output_recycling(bool(enabled))


def release_cached_outputs() -> None:
    """(extension) Drop the output tensors kept for re-use (see :func:`set_output_recycling`) and the chunk storage they pin."""
    if _host is not None:
        _host.release_recycled_outputs()


def _run(job: _Job) -> None:
    """Orchestration (worker thread or inline): plan, allocate, stage, enqueue, build views."""
    lib = _nat.lib() if (job.device.type == "cuda" or job.meta[0].max(initial=0) > R_REUSE) else None
    dev, tree = job.device, job.tree
    route, nbytes, esize, ptr, didx = job.meta
    to_gpu = dev.type == "cuda"

    for i in np.nonzero(route == R_OTHER)[0]:
        tree.set_out(int(i), tree.leaf(int(i)).to(dev))

    if to_gpu:
        side = _side_stream(dev)
        if job.packed is not None and job.packed.num_packed:
            with torch.cuda.device(dev):
                _enqueue_packed_batch(job, lib, side)
        packable = np.nonzero(route == R_H2D_PACK)[0].astype(np.int64)
        single = np.nonzero(route == R_H2D_SINGLE)[0].tolist()
        d2d = np.nonzero(route == R_D2D)[0].tolist()
        d2d_small = np.nonzero(route == R_D2D_SMALL)[0].astype(np.int64)
        for src_index in np.unique(didx[d2d_small]).tolist() if len(d2d_small) else []:
            group = d2d_small[didx[d2d_small] == src_index]
            if not (len(group) >= 2 and _coalesced_d2d(job, lib, group, int(src_index), side)):
                d2d = sorted(d2d + group.tolist())
        with torch.cuda.device(dev):
            prep = _prepare_packed_h2d(job, lib, packable, side) if len(packable) >= 2 else None
            if prep is not None:
                _nat.check(lib.accv_mtc_stage_h2d(*prep["stage_args"], side.cuda_stream, 0), "mtc_stage_h2d")
                # typed views into the chunk storage (enqueue_packed_transfer, multi_tensor_copier.cpp:712-729)
                tree.make_packed_views(*prep["views"])
            else:
                single = sorted(single + packable.tolist())
            # ---- everything else that targets the GPU goes through torch on the side stream
            if single or d2d:
                ready = torch.cuda.Event()
                ready.record(job.caller_stream)
                side.wait_event(ready)
                for ev in job.source_events.values():
                    side.wait_event(ev)           # synchronize_source_streams, multi_tensor_copier.cpp:741-762
                outs = {}
                with torch.cuda.stream(job.caller_stream):
                    for i in single + d2d:
                        outs[i] = torch.empty_like(tree.leaf(i), device=dev)
                with torch.cuda.stream(side):
                    for i in single:
                        src_t = tree.leaf(i)
                        if job.pinned and not src_t.is_pinned():
                            src_t = src_t.contiguous().pin_memory()
                            job.keep.append(src_t)
                        outs[i].copy_(src_t, non_blocking=True)
                    for i in d2d:
                        outs[i].copy_(tree.leaf(i), non_blocking=True)
                for i, o in outs.items():
                    tree.set_out(i, o)
            done = torch.cuda.Event()
            done.record(side)
            job.events.append(done)
    else:
        d2h_all = np.nonzero((route == R_D2H_SMALL) | (route == R_D2H_OTHER))[0]
        for dev_index in np.unique(didx[d2h_all]).tolist() if len(d2h_all) else []:
            sdev = torch.device("cuda", int(dev_index))
            side = _side_stream(sdev)
            on_dev = d2h_all[didx[d2h_all] == dev_index]
            small = on_dev[route[on_dev] == R_D2H_SMALL].astype(np.int64)
            with torch.cuda.device(sdev):
                side.wait_event(job.source_events[int(dev_index)])
                packed_ok = len(small) >= 2 and _coalesced_d2h(job, lib, small, sdev, side)
                rest = on_dev.tolist() if not packed_ok else on_dev[route[on_dev] == R_D2H_OTHER].tolist()
                with torch.cuda.stream(side):
                    for i in rest:
                        t = tree.leaf(i)
                        out = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=job.pinned)
                        out.copy_(t, non_blocking=job.pinned)
                        tree.set_out(i, out)
                done = torch.cuda.Event()
                done.record(side)
                job.events.append(done)


def _enqueue_packed_batch(job: _Job, lib, side) -> None:
    """Host->GPU for a PackedBatch: the packing already happened in the producer (DataLoader worker), so this is ONE
    transfer of its buffer — straight from the buffer when it is pinned, through one arena block otherwise — and the
    typed views on the GPU chunk."""
    pk, dev = job.packed, job.device
    size = pk.buffer.numel()
    align = pk.alignment
    with torch.cuda.stream(job.caller_stream):
        g = torch.empty(size + align + 15, dtype=torch.uint8, device=dev)
        base = (-g.data_ptr()) % align
        ready = torch.cuda.Event()
        ready.record(job.caller_stream)
    side.wait_event(ready)
    dst = g.data_ptr() + base
    if pk.buffer.is_pinned() or not job.pinned:
        _nat.check(lib.accv_memcpy_async(dst, pk.buffer.data_ptr(), size, 1, side.cuda_stream), "memcpy_async")
    else:
        stage = lib.accv_pinned_acquire(size)
        if not stage:
            _nat.check(-4, "pinned arena")
        job.staging.append(stage)
        # the worker pool splits at item boundaries: present the buffer as 1 MiB slices
        piece = 1 << 20
        m = (size + piece - 1) // piece
        off = np.arange(m, dtype=np.int64) * piece
        nb = np.minimum(piece, size - off).astype(np.int64)
        src = (np.uint64(pk.buffer.data_ptr()) + off.astype(np.uint64)).astype(np.uint64)
        order = np.arange(m, dtype=np.int64)
        begin = np.array([0, m], dtype=np.int64)
        one = lambda v, t: np.array([v], dtype=t)  # noqa: E731
        stage_a, dev_a, csz = one(stage, np.uint64), one(dst, np.uint64), one(size, np.int64)
        _nat.check(lib.accv_mtc_stage_h2d(m, src.ctypes.data, nb.ctypes.data, off.ctypes.data, order.ctypes.data, 1,
                                          begin.ctypes.data, stage_a.ctypes.data, dev_a.ctypes.data, csz.ctypes.data,
                                          side.cuda_stream, 0), "mtc_stage_h2d")
    job.tree.views_on(g, int(base), pk._leaf_ids, pk._offsets, pk._dtypes, pk._ndims, pk._shapes, False)
    job.keep.append(g)


def _item_table_to_device(job: _Job, lib, table: np.ndarray, sdev: torch.device, stream) -> torch.Tensor:
    """{pointer, offset, bytes} table of a coalescing launch -> device memory through a pinned arena block and ONE
    asynchronous copy on `stream` (a pageable .to(device) would block the host and serialise with the stream)."""
    tbytes = int(table.nbytes)
    block = lib.accv_pinned_acquire(tbytes)
    if not block:
        _nat.check(-4, "pinned arena")
    job.staging.append(block)                      # goes back to the arena once the job's events have completed
    ctypes.memmove(block, table.ctypes.data, tbytes)
    items = torch.empty(tbytes, dtype=torch.uint8, device=sdev)
    _nat.check(lib.accv_memcpy_async(items.data_ptr(), block, tbytes, 1, stream.cuda_stream), "memcpy_async")
    return items


def _coalesced_d2d(job: _Job, lib, group: np.ndarray, src_index: int, side_dst) -> bool:
    """Many small tensors of GPU `src_index` -> ONE gather kernel there (accv_mtc_coalesce) -> ONE device-to-device copy
    -> typed views of one storage on the target GPU (the reference copies every tensor on its own,
    multi_tensor_copier.cpp:775-820)."""
    route, nbytes, esize, ptr, _ = job.meta
    off, chk, csz, n_chunks, gbytes = _plan(lib, job, nbytes[group], esize[group])[:5]
    if n_chunks == 0:
        return False
    sdev, ddev = torch.device("cuda", src_index), job.device
    side_src = _side_stream(sdev)
    ptrs = np.ascontiguousarray(ptr[group], dtype=np.uint64)
    chunks, bases = [], np.zeros(n_chunks, dtype=np.int64)
    with torch.cuda.device(sdev):
        side_src.wait_event(job.source_events[src_index])
        gathered = []
        with torch.cuda.stream(side_src):
            for c in range(n_chunks):
                sel = np.nonzero(chk == c)[0]
                table = np.empty((len(sel), 3), dtype=np.int64)
                table[:, 0] = ptrs[sel].view(np.int64)
                table[:, 1] = off[sel]
                table[:, 2] = gbytes[sel]
                items = _item_table_to_device(job, lib, table, sdev, side_src)
                packed = torch.empty(int(csz[c]) + 16, dtype=torch.uint8, device=sdev)
                _nat.check(lib.accv_mtc_coalesce(items.data_ptr(), len(sel), packed.data_ptr(), 0, side_src.cuda_stream),
                           "mtc_coalesce")
                gathered.append(packed)
                job.keep.append((items, packed))
        src_done = torch.cuda.Event()
        src_done.record(side_src)
    with torch.cuda.device(ddev):
        with torch.cuda.stream(job.caller_stream):
            for c in range(n_chunks):
                chunks.append(torch.empty((int(csz[c]) + 31) // 16 * 16, dtype=torch.uint8, device=ddev))
            ready = torch.cuda.Event()
            ready.record(job.caller_stream)
        side_dst.wait_event(ready)
        side_dst.wait_event(src_done)
        with torch.cuda.stream(side_dst):
            for c in range(n_chunks):
                chunks[c][:int(csz[c]) + 16].copy_(gathered[c], non_blocking=True)
    job.keep.append(chunks)
    job.tree.make_packed_views(group, chk, off, chunks, bases)
    return True


def _coalesced_d2h(job: _Job, lib, small: np.ndarray, sdev: torch.device, side) -> bool:
    """Many small device tensors -> ONE device gather kernel (accv_mtc_coalesce) -> ONE D2H transfer per chunk -> host
    views (the reference copies each tensor separately, multi_tensor_copier.cpp:790-800)."""
    route, nbytes, esize, ptr, _ = job.meta
    off, chk, csz, n_chunks, sbytes = _plan(lib, job, nbytes[small], esize[small])[:5]
    if n_chunks == 0:
        return False
    ptrs = np.ascontiguousarray(ptr[small], dtype=np.uint64)
    hosts, bases = [], np.zeros(n_chunks, dtype=np.int64)
    with torch.cuda.stream(side):
        for c in range(n_chunks):
            sel = np.nonzero(chk == c)[0]
            size = int(csz[c])
            table = np.empty((len(sel), 3), dtype=np.int64)      # {src pointer, offset in packed, nbytes}
            table[:, 0] = ptrs[sel].view(np.int64)
            table[:, 1] = off[sel]
            table[:, 2] = sbytes[sel]
            items = _item_table_to_device(job, lib, table, sdev, side)
            packed = torch.empty(size + 16, dtype=torch.uint8, device=sdev)
            _nat.check(lib.accv_mtc_coalesce(items.data_ptr(), len(sel), packed.data_ptr(), 0, side.cuda_stream),
                       "mtc_coalesce")
            host = torch.empty((size + 31) // 16 * 16, dtype=torch.uint8, device="cpu", pin_memory=job.pinned)
            host[:size + 16].copy_(packed, non_blocking=job.pinned)
            job.keep.append((items, packed))
            hosts.append(host)
    job.tree.make_packed_views(small, chk, off, hosts, bases)
    return True


def _abandon(job: _Job) -> None:
    """A copy failed half way: wait for whatever was already enqueued on the side streams (the completion event may
    never have been recorded), then hand the staging blocks back."""
    try:
        if job.ticket is not None:          # let the native job finish before its staging blocks are recycled
            ticket, job.ticket = job.ticket, None
            _nat.ctypes_lib().accv_mtc_async_wait(ticket)
        job.pending_views = None
        for ev in job.events:
            ev.synchronize()
        devs = set(job.source_events)
        if job.device.type == "cuda":
            devs.add(job.device.index)
        for i in devs:
            side = _side_streams.get(int(i))
            if side is not None:
                side.synchronize()
    finally:
        job.release_staging()


def _rebuild_without_gc(tree):
    """Build the output structure with the cyclic collector paused.  A 10k-leaf result is 12k fresh GC-tracked
    objects; with the collector running, the generation-0 passes it triggers every 700 allocations promote the
    half-built result generation by generation and regularly end in a FULL collection of the process heap (25 ms in a
    torch process; 60 % of the time of a 10k-tensor copy, profiles/r01_mtc_breakdown.log).  Nothing allocated here can
    be garbage before this function returns, so pausing loses nothing; the previous collector state is restored."""
    if tree.num_leaves() < 256 or not gc.isenabled():
        return tree.rebuild()
    gc.disable()
    try:
        return tree.rebuild()
    finally:
        gc.enable()


class AsyncCopyHandle:
    """Handle to an in-progress copy started by :func:`start_copy`.  ``ready()`` polls, ``get()`` blocks and returns
    the copied structure.  Dropping the handle early blocks in the destructor until the transfers are done so that
    staging buffers are never recycled under an active DMA (reference: multi_tensor_copier.cpp:932-943)."""

    def __init__(self, job: _Job, future: Optional[Future]):
        self._job = job
        self._future = future
        self._result = None
        self._consumed = False
        self._error: Optional[BaseException] = None

    def _worker_result(self) -> None:
        """Re-raise a worker exception — after the transfers that were already enqueued have drained and the arena
        blocks acquired before the failure went back (otherwise they would leak for the life of the process, and the
        keep-alives of an active DMA would be dropped with the handle)."""
        job = self._job
        if self._error is not None:              # a failed copy keeps failing (the reference stores and rethrows it)
            raise self._error
        try:
            if job.pending_views is not None:
                # native orchestration: the typed views only need the plan and the chunks, so they are built here, on the
                # consumer's thread, WHILE the library thread stages and enqueues
                views, job.pending_views = job.pending_views, None
                job.tree.make_packed_views(*views)
            if job.ticket is not None:
                ticket, job.ticket = job.ticket, None
                _nat.check(_nat.lib().accv_mtc_async_wait(ticket), "mtc_async_wait")   # ctypes: lock released
                done = torch.cuda.Event()
                done.record(job.side)          # every transfer of the job is on the side stream by now
                job.events.append(done)
            if self._future is not None:
                future, self._future = self._future, None
                if future.cancel():
                    # the worker has not picked the job up yet (`start_copy(...).get()` right away): run it here instead of
                    # waiting for a thread hand-off and the interpreter-lock ping-pong that follows it (10 000 GPU->host
                    # leaves: 1.9 -> 1.0 ms); a job that is already running is waited for as before
                    _run(job)
                else:
                    future.result()
        except BaseException as e:
            self._error = e
            _abandon(job)
            raise

    def _finish(self) -> None:
        self._worker_result()
        try:
            for ev in self._job.events:
                ev.synchronize()
        finally:
            self._job.release_staging()

    def ready(self) -> bool:
        """True once the copy has completed (non-blocking).  Raises if the copy failed."""
        if self._future is not None and not self._future.done():
            return False
        if self._job.ticket is not None and _nat.lib().accv_mtc_async_poll(self._job.ticket) == 0:
            return False
        self._worker_result()
        if all(ev.query() for ev in self._job.events):
            self._job.release_staging()
            return True
        return False

    def get(self):
        """Block until done; return the input structure with every tensor on the target device."""
        if not self._consumed:
            self._finish()
            self._result = _rebuild_without_gc(self._job.tree)
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
            other leaves (the latter are passed through unchanged; other container types count as opaque leaves);
            or (extension) a :class:`PackedBatch` made by ``pack_batch`` / ``packing_collate`` in a DataLoader worker.
        device: target device (``"cuda:0"``, ``"cpu"``, ``torch.device``).
        use_pinned_staging: stage host<->GPU transfers through pinned memory (host->GPU becomes asynchronous;
            GPU->host returns the pinned buffer itself).
        pack_cpu_tensors: coalesce small contiguous tensors (<= 256 KiB each, any dtype mix) into shared chunks with one
            transfer per chunk (host->GPU as in the reference; also applied to GPU->host here).
        min_packed_alignment_bytes: each packed tensor starts at a multiple of
            ``round_up(max(min_packed_alignment_bytes, element_size), element_size)`` inside its chunk.
        max_packed_chunk_bytes: payload limit per chunk (default 32 MiB).
        use_background_thread: run allocation, staging and enqueueing on a pool thread so this call returns early.

    The inputs must stay alive and unmodified until ``get()`` returned or ``ready()`` was True.
    """
    try:
        dev = torch.device(device)
    except RuntimeError as exc:
        # the extension's own text for a malformed device (multi_tensor_copier.cpp:225-232: parse_device turns the
        # c10::Error into this message); torch's explanation stays attached as the cause
        raise RuntimeError(f"Invalid device string: '{device}'") from exc
    if dev.type == "cuda":
        if not torch.cuda.is_available():
            raise RuntimeError(f"Invalid device string: '{device}' (no GPU is available)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        _nat.lib()  # fail loudly if the HIP library is missing
    elif dev.type != "cpu":
        raise RuntimeError(f"Invalid device string: '{device}'")
    from .packed import PackedBatch

    packed = data if isinstance(data, PackedBatch) else None
    if packed is not None:
        # produced by pack_batch / packing_collate: the small tensors already sit in one buffer
        tree = packed._tree(cpu_views=dev.type != "cuda")
    else:
        tree = _make_leaf_set(data)
    job = _Job(tree, dev, bool(use_pinned_staging), bool(pack_cpu_tensors), int(min_packed_alignment_bytes),
               int(max_packed_chunk_bytes))
    if packed is not None and dev.type == "cuda":
        job.packed = packed              # also keeps the buffer alive until the handle is consumed
    job.meta = tree.classify(str(dev), job.pack)
    route, _, _, _, didx = job.meta
    # ordering: capture the caller's current streams NOW (reference :1086-1123)
    if dev.type == "cuda":
        job.caller_stream = torch.cuda.current_stream(dev)
    if os.environ.get("ACCV_MTC_D2D_SAME_DEVICE") == "1" and dev.type == "cuda":
        # test hook for one-GPU boxes: treat small tensors that already sit on the target device as if they came from
        # another GPU, so that the coalesced GPU->GPU path runs (gather kernel + one copy + views) — results are copies
        small = (route == R_REUSE) & (nbytes_all(job) > 0) & (nbytes_all(job) <= PACK_MAX_BYTES_PER_TENSOR) & (didx == dev.index)
        route[small] = R_D2D_SMALL
    for dev_index in np.unique(didx[((route >= R_D2H_SMALL) & (route <= R_D2D)) | (route == R_D2D_SMALL)]).tolist():
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(torch.device("cuda", int(dev_index))))
        job.source_events[int(dev_index)] = ev
    if use_background_thread:
        try:
            if _start_native_h2d(job):          # native library thread: no python worker, no interpreter-lock hand-offs
                return AsyncCopyHandle(job, None)
        except Exception as e:                  # background mode reports failures from ready() / get(), as a worker would
            _abandon(job)
            failed = AsyncCopyHandle(job, None)
            failed._error = e
            return failed
        return AsyncCopyHandle(job, _executor().submit(_run, job))
    try:
        _run(job)  # inline: exceptions propagate from start_copy (reference :1151-1153)
    except BaseException:
        _abandon(job)
        raise
    return AsyncCopyHandle(job, None)
