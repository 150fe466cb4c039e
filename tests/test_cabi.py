"""CPU-only checks of the drop-in boundary: libaccv_hip.so loads and exports exactly the symbols that
include/*.h declare; argument validation that needs no GPU returns the documented status codes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            text = open(os.path.join(inc, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(accv_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    from accvlab import _amd_native as nat

    assert os.path.exists(nat.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(nat.LIB_PATH)
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/ but not exported"
    # and the python binding table covers every declared symbol
    assert declared == set(nat.SIGNATURES), declared ^ set(nat.SIGNATURES)


def test_version_and_error_string():
    from accvlab import _amd_native as nat

    lib = nat.lib()
    assert lib.accv_version() >= 100
    assert isinstance(lib.accv_last_error(), (bytes, type(None)))


@pytest.mark.parametrize("path", ["trampoline", "ctypes"])
def test_argument_validation_without_gpu(path):
    """the same entry points through both host bindings: the METH_FASTCALL trampoline (accvlab/_amd_native/_fastcall,
    what the operators use) and plain ctypes"""
    from accvlab import _amd_native as nat

    lib = nat.lib() if path == "trampoline" else nat.ctypes_lib()
    if path == "trampoline":
        assert nat._fastcall is not None, "build the host extensions (make -C accv-lab_amd/csrc_host)"
        import functools
        assert isinstance(lib.accv_ragged_gather, functools.partial)          # int-only signature -> call_ints
    # negative extents -> ACCV_EINVAL before anything touches the device
    assert lib.accv_draw_heatmap_batched_f32(None, 1, 0, -1, 4, None, None, None, None, 0, 6.0, 1.0, 0, None) == -1
    assert b"negative" in lib.accv_last_error()
    # labels without classes
    dummy = 16 if path == "trampoline" else ctypes.c_void_p(16)
    assert lib.accv_draw_heatmap_batched_f32(dummy, 1, 0, 4, 4, dummy, dummy, dummy, dummy, 1, 6.0, 1.0, 0, None) == -1
    # workspace too small
    assert lib.accv_draw_heatmap_flat_f32(dummy, 2, 4, 4, dummy, dummy, dummy, 3, 6.0, 1.0, 0, dummy, 8, None) == -3
    assert lib.accv_draw_heatmap_flat_workspace_bytes(2, 3) >= (2 + 3 + 3) * 4
    # empty problems succeed without touching the device
    assert lib.accv_draw_heatmap_batched_f32(None, 0, 0, 4, 4, None, None, None, None, 0, 6.0, 1.0, 0, None) == 0
    assert lib.accv_fill_f32(None, 0, 0.0, None) == 0
    # integer-class arguments: negative values, None pointers and the error string survive the trampoline
    assert lib.accv_ragged_gather(dummy, dummy, dummy, dummy, -1, 4, 4, 4, 16, 0, 0, None, None) == -1
    assert b"invalid extents" in lib.accv_last_error()
    assert lib.accv_ragged_gather(None, None, None, None, 0, 4, 4, 4, 16, 0, 0, None, None) == 0
    with pytest.raises(nat.AccvNativeError):
        nat.check(-1, "unit test")


def test_lane_raster_shape_rule_and_rider_argument_checks_without_gpu():
    """host logic of the round-3 lane entry points: which shapes the one-launch lane raster takes
    (accv_draw_polylines_fused_applicable) and what the sampler rider refuses — decided before anything touches the device"""
    from accvlab import _amd_native as nat

    lib = nat.ctypes_lib()
    one = lambda *v: (ctypes.c_int * len(v))(*v)
    hs, ws = one(512), one(1024)
    # 64 point slots per frame: points rounded up to a power of two (>= 4) x polylines
    for lanes, points, samples, want in ((1, 64, 32, 1), (2, 64, 32, 0), (1, 65, 32, 0), (16, 4, 6, 1), (17, 4, 6, 0), (16, 4, 7, 0),
                                         (4, 16, 32, 1), (5, 16, 32, 0), (2, 17, 32, 1), (3, 17, 32, 0), (2, 24, 368, 1), (2, 24, 369, 0),
                                         (0, 24, 32, 0), (2, 0, 32, 0), (2, 24, 0, 0)):
        assert lib.accv_draw_polylines_fused_applicable(hs, ws, 1, 2, lanes, points, samples) == want, (lanes, points, samples)
    assert lib.accv_draw_polylines_fused_applicable(None, ws, 1, 2, 1, 24, 32) == 0
    assert lib.accv_draw_polylines_fused_applicable(hs, ws, 5, 2, 1, 24, 32) == 0               # more than four scales
    assert lib.accv_draw_polylines_fused_applicable(one(16), one(128), 1, 2, 2, 24, 256) == 0   # coarse tiles in the majority
    assert lib.accv_draw_polylines_fused_applicable(one(256), one(1024), 1, 2, 2, 24, 256) == 1
    # the fused entry point refuses shapes outside that rule, the rider polylines the wave-level sampler does not take
    d = ctypes.c_void_p(64)
    st = (ctypes.c_float * 1)(1.0)
    ptrs = (ctypes.c_void_p * 1)(64)
    assert lib.accv_draw_polylines_multiscale_f32(ptrs, hs, ws, st, 1, 2, d, 5, 64, None, d, 96, 2, 6.0, 1.0, 1, None) == -1
    assert b"fused" in lib.accv_last_error()
    assert lib.accv_draw_polylines_multiscale_f32(ptrs, hs, ws, st, 1, 0, d, 5, 64, None, d, 96, 2, 6.0, 1.0, 1, None) == 0     # empty batch
    rider = lambda points, samples, out: lib.accv_draw_heatmap_multiscale_sample_f32(
        ptrs, hs, ws, st, 1, 2, None, None, d, 0, 6.0, 1.0, 0, d, 6, points, None, samples, out, d, None)
    assert rider(65, 64, d) == -1 and b"64 points" in lib.accv_last_error()
    assert rider(0, 64, d) == -1
    assert rider(9, 100, d) == -1 and b"multiple of 64" in lib.accv_last_error()
    assert rider(9, 64, None) == -1 and b"null" in lib.accv_last_error()
    assert rider(9, 64, ctypes.c_void_p(68)) == -1 and b"alignment" in lib.accv_last_error()


@pytest.mark.parametrize("value", [4, 1, 0])
def test_null_pointer_sweep_is_rejected_or_empty(value):
    """every int-returning entry point called with NULL for every pointer and `value` for every integer: the call must come
    back with ACCV_OK (an empty problem / nothing requested), ACCV_EINVAL or ACCV_EWORKSPACE — never crash, and never get as far
    as a kernel launch (ACCV_ELAUNCH).  accv_memcpy_async is a bare hipMemcpyAsync and reports the runtime's own error."""
    from accvlab import _amd_native as nat

    lib = nat.ctypes_lib()
    for name, (res, args) in sorted(nat.SIGNATURES.items()):
        if res is not ctypes.c_int or not args:
            continue
        vals = [None if a in (ctypes.c_void_p, ctypes.c_char_p) else 1.0 if a in (ctypes.c_float, ctypes.c_double) else value
                for a in args]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
        status = fn(*vals)
        allowed = (0, -1, -3, -2) if name == "accv_memcpy_async" else (0, -1, -3)
        assert status in allowed, f"{name}({value}): status {status}: {lib.accv_last_error()}"


@pytest.mark.parametrize("value", [3, 0, -2])
def test_every_trampoline_eligible_entry_point_behaves_like_its_ctypes_binding(value):
    """the trampoline calls every entry point through one fixed 20-slot function type.  Each entry point it is
    eligible for is therefore called through BOTH bindings with the same arguments (NULL pointers, one integer value, 1.5 /
    2.5 for the floats): the status and the thread's error text must be identical — a mis-passed argument changes which
    validation fires first.  (The GPU suite additionally runs once over ctypes alone: profiles/r03_gpu_tests_ctypes_binding.log.)"""
    from accvlab import _amd_native as nat

    assert nat._fastcall is not None, "build the host extensions (make -C accv-lab_amd/csrc_host)"
    plain = nat.ctypes_lib()
    checked = 0
    for name, (res, args) in sorted(nat.SIGNATURES.items()):
        if name in nat._BLOCKING or not args:
            continue
        fast = nat._fast_entry(getattr(plain, name), res, args)
        if fast is None:
            continue
        floats = iter((1.5, 2.5))
        vals = [None if a in (ctypes.c_void_p, ctypes.c_char_p) else next(floats) if a is ctypes.c_float else value for a in args]
        s_fast, e_fast = fast(*vals), plain.accv_last_error()
        s_plain, e_plain = getattr(plain, name)(*vals), plain.accv_last_error()
        assert s_fast == s_plain, f"{name}({value}): trampoline {s_fast} vs ctypes {s_plain}"
        assert s_fast == 0 or e_fast == e_plain, f"{name}({value}): {e_fast!r} vs {e_plain!r}"
        checked += 1
    assert checked >= 20, f"only {checked} entry points went through the trampoline"


def test_trampoline_passes_values_and_pointers_like_ctypes():
    """a call with real pointers, large and negative integers: the pack planner fills the same arrays through both bindings"""
    import numpy as np

    from accvlab import _amd_native as nat

    plain = nat.ctypes_lib()
    fast = nat._fast_entry(plain.accv_mtc_plan, *nat.SIGNATURES["accv_mtc_plan"])
    assert fast is not None
    rng = np.random.default_rng(0)
    n = 300
    nbytes = rng.integers(1, 5000, n).astype(np.int64)
    esize = rng.choice([1, 2, 4, 8, 16], n).astype(np.int32)
    cand = (rng.random(n) < 0.8).astype(np.uint8)
    outs = []
    for call in (fast, plain.accv_mtc_plan):
        off, chk, csz = np.full(n, -9, np.int64), np.full(n, -9, np.int64), np.full(n, -9, np.int64)
        k = ctypes.c_longlong(-1)
        args = (n, nbytes.ctypes.data, esize.ctypes.data, cand.ctypes.data, 16, 1 << 40, off.ctypes.data, chk.ctypes.data,
                csz.ctypes.data, ctypes.addressof(k))
        assert call(*args) == 0
        outs.append((off.copy(), chk.copy(), csz[:k.value].copy(), k.value))
    assert outs[0][3] == outs[1][3] >= 1
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a, b)
