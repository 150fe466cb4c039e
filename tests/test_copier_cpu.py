"""CPU-only checks for H3: the native pack planner against the oracle and the reference-derived offsets, and the
host->host paths of start_copy (no GPU involved)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import h3 as oracle


def native_plan(nbytes, esizes, cand, min_align, max_chunk):
    from accvlab import _amd_native as nat

    n = len(nbytes)
    nb = np.asarray(nbytes, dtype=np.int64)
    es = np.asarray(esizes, dtype=np.int32)
    cd = np.asarray(cand, dtype=np.uint8)
    off = np.empty(n, dtype=np.int64)
    chk = np.empty(n, dtype=np.int64)
    csz = np.empty(max(n, 1), dtype=np.int64)
    k = ctypes.c_longlong(0)
    rc = nat.lib().accv_mtc_plan(n, nb.ctypes.data, es.ctypes.data, cd.ctypes.data, min_align, max_chunk,
                                 off.ctypes.data, chk.ctypes.data, csz.ctypes.data, ctypes.addressof(k))
    assert rc == 0
    return off.tolist(), chk.tolist(), csz[:k.value].tolist()


# the five packable leaves of the reference test, in traversal order: f32[8,4], i64[17], f16[11], c64[9], c128[5]
REF_LEAVES = ([128, 136, 22, 72, 80], [4, 8, 2, 8, 16])


@pytest.mark.parametrize("min_align,offsets,chunk", [
    (16, [0, 128, 272, 304, 384], 464),
    (1, [288, 80, 416, 216, 0], 438),      # layout order c128, i64, c64, f32, f16
    (6, [80, 208, 420, 344, 0], 442),      # c128@0, f32@80, i64@208, c64@344, f16@420
])
def test_plan_reference_derived_offsets(min_align, offsets, chunk):
    nb, es = REF_LEAVES
    for fn in (oracle.plan, native_plan):
        off, chk, sizes = fn(nb, es, [1] * 5, min_align, 32 << 20)
        assert off == offsets and chk == [0] * 5 and sizes == [chunk], fn
    for o, e in zip(offsets, es):
        assert o % oracle.required_align(min_align, e) == 0


def test_plan_random_native_equals_oracle_and_invariants():
    g = np.random.RandomState(0)
    for trial in range(200):
        n = int(g.randint(0, 60))
        es = g.choice([1, 2, 4, 8, 16], n).tolist()
        nb = [int(e * g.randint(1, 3000)) for e in es]
        cand = (g.rand(n) > 0.2).astype(int).tolist()
        min_align = int(g.choice([1, 2, 6, 16, 24, 64, 100]))
        max_chunk = int(g.choice([1 << 12, 1 << 15, 1 << 25]))
        a = oracle.plan(nb, es, cand, min_align, max_chunk)
        b = native_plan(nb, es, cand, min_align, max_chunk)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], trial
        off, chk, sizes = b
        spans = {}
        for i in range(n):
            if off[i] < 0:
                continue
            assert cand[i] and off[i] % oracle.required_align(min_align, es[i]) == 0
            assert off[i] + nb[i] <= sizes[chk[i]]
            spans.setdefault(chk[i], []).append((off[i], off[i] + nb[i]))
        for c, sp in spans.items():
            sp.sort()
            assert all(x[1] <= y[0] for x, y in zip(sp, sp[1:])), "overlap"
            assert sizes[c] <= max_chunk or len(sp) == 1
        if sum(1 for o in off if o >= 0) == 1:
            pytest.fail("a single tensor must not be packed")


def test_plan_fewer_than_two_candidates_disables_packing():
    assert native_plan([100], [4], [1], 16, 1 << 20) == ([-1], [-1], [])
    assert native_plan([100, 50], [4, 2], [1, 0], 16, 1 << 20) == ([-1, -1], [-1, -1], [])
    assert native_plan([], [], [], 16, 1 << 20) == ([], [], [])


def test_start_copy_numpy_to_cpu_and_passthrough():
    """reference behaviours: tests/test_multi_tensor_copier.py:73-97 (numpy -> CPU tensors) and :100-125 (passthrough
    identity, container types)."""
    import accvlab.multi_tensor_copier as mtc

    marker = object()
    data = [np.arange(12, dtype=np.float32).reshape(3, 4),
            (np.ones((2, 3), dtype=np.float32), [np.zeros((1,), dtype=np.int64)]),
            {"k": torch.arange(3), "m": marker, "s": "text", 7: None}]
    for bg in (True, False):
        h = mtc.start_copy(data, "cpu", pack_cpu_tensors=False, use_background_thread=bg)
        out = h.get()
        assert h.ready() is True
        assert isinstance(out, list) and isinstance(out[1], tuple) and isinstance(out[1][1], list)
        assert isinstance(out[0], torch.Tensor) and out[0].device.type == "cpu"
        torch.testing.assert_close(out[0], torch.from_numpy(data[0]))
        torch.testing.assert_close(out[1][1][0], torch.from_numpy(data[1][1][0]))
        assert out[2]["m"] is marker and out[2]["s"] == "text" and out[2][7] is None
        assert out[2]["k"] is data[2]["k"]          # already on the target device: reused as is
        assert list(out[2].keys()) == ["k", "m", "s", 7]
    single = mtc.start_copy(torch.arange(4), "cpu").get()
    assert isinstance(single, torch.Tensor)
    with pytest.raises(RuntimeError):
        mtc.start_copy([torch.zeros(1)], "not-a-device")


def test_invalid_device_string_message():
    """A malformed device surfaces as RuntimeError("Invalid device string: '…'") — the text the reference extension
    produces (multi_tensor_copier.cpp:225-232) — with torch's own explanation attached as the cause."""
    from accvlab.multi_tensor_copier import start_copy

    for bad in ("notadevice", "cuda:x", "gpu0"):
        with pytest.raises(RuntimeError, match=f"^Invalid device string: '{bad}'") as info:
            start_copy([torch.zeros(2)], bad)
        assert info.value.__cause__ is not None
    with pytest.raises(RuntimeError, match="^Invalid device string: 'meta'"):      # a device type this path does not serve
        start_copy([torch.zeros(2)], "meta")


# ---- life cycle of the library's orchestration thread: bounded ticket table, fork, orderly shutdown.
# Staging-only jobs (device pointer NULL = no transfer) need no GPU; each case runs in a child interpreter because the
# shutdown is final for a process.
_LIFECYCLE = r'''
import ctypes, os, sys
import numpy as np
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]
from accvlab import _amd_native as nat
lib = nat.ctypes_lib()

def stage(n=3, nbytes=64):
    src = [np.arange(nbytes, dtype=np.uint8) + k for k in range(n)]
    dst = np.zeros(n * nbytes, dtype=np.uint8)
    P, L = ctypes.c_void_p, ctypes.c_longlong
    a_src = (P * n)(*[s.ctypes.data for s in src])
    a_nb = (L * n)(*([nbytes] * n)); a_off = (L * n)(*[k * nbytes for k in range(n)]); a_ord = (L * n)(*range(n))
    a_beg = (L * 2)(0, n); a_stg = (P * 1)(dst.ctypes.data); a_dev = (P * 1)(None); a_cb = (L * 1)(n * nbytes)
    t = L(0)
    rc = lib.accv_mtc_stage_h2d_async(n, a_src, a_nb, a_off, a_ord, 1, a_beg, a_stg, a_dev, a_cb, None, 1, -1, ctypes.addressof(t))
    return rc, t.value, dst, src
'''


def _run_child(body):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"ROOT = {root!r}\n" + _LIFECYCLE + body
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    return res.stdout


def test_orchestrator_forgets_finished_tickets_nobody_waited_for():
    out = _run_child(r'''
import time
for i in range(1500):                      # never waited for: only polled
    rc, t, dst, src = stage()
    assert rc == 0
    while lib.accv_mtc_async_poll(t) == 0:
        time.sleep(0)
held = lib.accv_mtc_async_tickets_held()
assert held <= 1024 + 1, held              # bounded, not 1500
rc, t, dst, src = stage()
assert lib.accv_mtc_async_wait(t) == 0 and bytes(dst[:64]) == bytes(src[0]) and bytes(dst[128:]) == bytes(src[2])
assert lib.accv_mtc_async_wait(t) != 0     # a ticket is forgotten by the wait that returned its status
assert lib.accv_mtc_async_wait(1) != 0     # swept long ago
print("ok")
''')
    assert "ok" in out


def test_orchestrator_works_in_a_forked_child_and_refuses_jobs_after_shutdown():
    out = _run_child(r'''
rc, t, dst, src = stage()
assert rc == 0 and lib.accv_mtc_async_wait(t) == 0          # the parent's thread exists now
pid = os.fork()
if pid == 0:                                                 # child: the parent's thread is gone, a fresh one must start
    import signal
    signal.alarm(30)                                         # a hang (the bug this guards against) ends the child
    rc, t, dst, src = stage()
    ok = rc == 0 and lib.accv_mtc_async_wait(t) == 0 and bytes(dst[64:128]) == bytes(src[1])
    os._exit(0 if ok else 1)
_, status = os.waitpid(pid, 0)
assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
rc, t, dst, src = stage()
assert rc == 0
lib.accv_mtc_shutdown()                                      # drains the queued job first
assert lib.accv_mtc_async_poll(t) == 1 and lib.accv_mtc_async_wait(t) == 0 and bytes(dst[:64]) == bytes(src[0])
lib.accv_mtc_shutdown()                                      # idempotent
rc, t, dst, src = stage()
assert rc != 0 and b"shut down" in lib.accv_last_error()
print("ok")
''')
    assert "ok" in out


# ---- output recycling of the C++ host path (round 3): driven directly on CPU "chunks", no GPU involved
def _views_of(host, tree_obj, chunk, offsets):
    t = host.Tree(tree_obj)
    n = t.num_leaves()
    t.make_packed_views(np.arange(n, dtype=np.int64), np.zeros(n, dtype=np.int64), offsets, [chunk], np.zeros(1, dtype=np.int64))
    return t.rebuild()


def test_output_tensor_objects_are_recycled_only_when_nothing_refers_to_them():
    import gc

    from accvlab.multi_tensor_copier import copier

    host = copier._host
    assert host is not None, "build the host extensions (make -C accv-lab_amd/csrc_host)"
    copier.release_cached_outputs()
    copier.set_output_recycling(True)
    leaves = [torch.zeros(4), torch.zeros(3, dtype=torch.int64), torch.zeros(2, 2)]
    offsets = np.array([0, 16, 48], dtype=np.int64)

    def result(fill):
        chunk = torch.zeros(80, dtype=torch.uint8)
        chunk[0:16] = torch.full((4,), float(fill)).view(torch.uint8)
        chunk[16:40] = torch.full((3,), int(fill), dtype=torch.int64).view(torch.uint8)
        chunk[48:64] = torch.full((4,), float(-fill)).view(torch.uint8)
        return _views_of(host, {"a": leaves[0], "b": [leaves[1], leaves[2]]}, chunk, offsets)

    r1 = result(1)
    kept = r1["a"]                       # the caller keeps ONE tensor of the first result (and a view of another)
    view = r1["b"][1][0]
    ids1 = (id(r1["a"]), id(r1["b"][0]), id(r1["b"][1]))
    del r1
    r2 = result(2)
    del r2
    gc.collect()
    r3 = result(3)                       # the tensors of result 1 that nobody holds may come back, re-pointed
    assert r3["a"].tolist() == [3.0] * 4 and r3["b"][0].tolist() == [3] * 3 and r3["b"][1].tolist() == [[-3.0, -3.0], [-3.0, -3.0]]
    assert kept.tolist() == [1.0] * 4, "a tensor the caller still holds was re-pointed"
    assert view.tolist() == [-1.0, -1.0], "a view of an old output lost its base"
    assert id(r3["a"]) != ids1[0] and id(r3["b"][1]) != ids1[2]          # held / viewed -> fresh objects
    # steady state: the objects alternate between two generations, none is created or destroyed
    seen = set()
    for k in range(4, 12):
        r = result(k)
        assert r["a"].tolist() == [float(k)] * 4 and r["b"][0].tolist() == [k] * 3
        seen.add((id(r["a"]), id(r["b"][0]), id(r["b"][1])))
        del r
    assert len(seen) <= 2, seen
    # an output the caller changed as an OBJECT (autograd flag, python attribute) is not handed out again
    for k in (30, 31):
        r = result(k)
        r["a"].requires_grad_(True)
        r["b"][0].note = "mine"
        del r
    for k in (32, 33, 34):
        r = result(k)
        assert not r["a"].requires_grad and not hasattr(r["b"][0], "note") and r["a"].tolist() == [float(k)] * 4
        del r
    # a different structure at the same positions: dtype mismatch -> fresh tensors, correct values
    chunk = torch.arange(80, dtype=torch.uint8)
    other = _views_of(host, [torch.zeros(16, dtype=torch.uint8), torch.zeros(6, dtype=torch.int32)], chunk, np.array([0, 16], dtype=np.int64))
    assert other[0].dtype == torch.uint8 and other[0].tolist() == list(range(16)) and other[1].dtype == torch.int32
    # switching it off / releasing empties the pool
    assert host.recycled_output_count() > 0
    copier.release_cached_outputs()
    assert host.recycled_output_count() == 0
    copier.set_output_recycling(False)
    a = result(20)
    ida = id(a["a"])
    del a
    b = result(21)
    assert host.recycled_output_count() == 0 and b["a"].tolist() == [21.0] * 4
    copier.set_output_recycling(True)


def test_background_job_runs_exactly_once_whether_get_comes_early_or_late(monkeypatch):
    """get() on a job the worker pool has not started yet takes it over and runs it on the caller's thread; a job that is
    running or done is waited for — either way the orchestration runs once and the result is right"""
    import threading
    import time

    from accvlab.multi_tensor_copier import copier, start_copy

    calls, threads = [], []
    real_run = copier._run

    def counting_run(job):
        calls.append(1)
        threads.append(threading.current_thread().name)
        return real_run(job)

    monkeypatch.setattr(copier, "_run", counting_run)
    data = {"a": [torch.arange(5), torch.ones(2, 3)], "b": (np.arange(4, dtype=np.float32), "tag")}

    def check(res):
        assert torch.equal(res["a"][0], torch.arange(5)) and torch.equal(res["a"][1], torch.ones(2, 3))
        assert torch.equal(res["b"][0], torch.arange(4, dtype=torch.float32)) and res["b"][1] == "tag"

    for wait in (0.0, 0.2):
        calls.clear()
        h = start_copy(data, "cpu", use_background_thread=True)
        if wait:
            time.sleep(wait)            # the worker has long finished
        check(h.get())
        check(h.get())                  # a second get() returns the same result without running anything
        assert len(calls) == 1, (wait, calls, threads)
    # ready() polling until done, then get(): still once
    calls.clear()
    h = start_copy(data, "cpu", use_background_thread=True)
    t0 = time.time()
    while not h.ready():
        assert time.time() - t0 < 10
        time.sleep(0.001)
    check(h.get())
    assert len(calls) == 1
