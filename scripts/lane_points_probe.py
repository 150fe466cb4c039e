#!/usr/bin/env python3
"""Where does splat_points_multi_kernel (the lane splat of BASELINE config 3) spend its time?  Calls the C-ABI entry point
directly on pre-sampled lanes (group boxes given) and varies: no lanes at all (empty-tile floor), the number of lanes, the
radius, the scales.  `--alt-lib PATH` times a second build of the library (scripts/build_prev_lib.sh) beside the shipped one
in the same process and checks that both write the same maps."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.draw_heatmap import sample_lanes  # noqa: E402


def gpu_us(fn, n=200):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    lib = nat.lib()
    libs = {"shipped": lib}
    for path in [a for a in sys.argv[1:] if a.endswith(".so")]:      # --alt-lib a.so b.so ...: further builds, same process
        alt = ctypes.CDLL(os.path.abspath(path))
        alt.accv_draw_points_multiscale_f32.restype, alt.accv_draw_points_multiscale_f32.argtypes = \
            nat.SIGNATURES["accv_draw_points_multiscale_f32"]
        libs[os.path.basename(path).replace("libaccv_hip_", "").replace(".so", "")] = alt
    B, SH, SW, L, P, Q = 32, 2160, 3840, 8, 24, 256
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    lanes = torch.stack([xs, ys], -1).to(dev)
    n = L * Q
    ws_bytes = lib.accv_draw_points_workspace_bytes(B, n)
    work = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    samples = sample_lanes(lanes, Q, group_boxes_ptr=work.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream

    def run(strides, counts, radius, clear=True):
        maps = [torch.empty(B, int(SH / s), int(SW / s), device=dev) for s in strides]
        k = len(maps)
        ptrs = (ctypes.c_void_p * k)(*[m.data_ptr() for m in maps])
        hs = (ctypes.c_int * k)(*[m.size(1) for m in maps])
        ws_ = (ctypes.c_int * k)(*[m.size(2) for m in maps])
        st = (ctypes.c_float * k)(*strides)
        flags = (nat.HM_CLEAR if clear else 0) | nat.HM_GROUP_BOXES_GIVEN
        nbytes = sum(m.numel() * 4 for m in maps)
        out, ref = {}, None
        for name, handle in libs.items():
            fn = lambda: nat.check(handle.accv_draw_points_multiscale_f32(ptrs, hs, ws_, st, k, B, samples.data_ptr(), counts.data_ptr(),
                                                                          n, radius, 6.0, 1.0, flags, work.data_ptr(), ws_bytes, stream), "points")
            for m in maps:
                m.fill_(0.25)
            us = gpu_us(fn)
            out[name] = {"us": round(us, 1), "GBps": round(nbytes / us / 1e3, 1)}
            snap = [m.clone() for m in maps]
            if ref is None:
                ref = snap
            else:
                out[name]["same_as_shipped"] = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, snap))
        out["kernel"] = nat.last_dispatch()
        return out

    full = torch.full((B,), n, dtype=torch.int32, device=dev)
    if "--rule" in sys.argv:
        # A/B build only: is the waves-per-tile rule (four waves only when coarse tiles are half of the launch) right for other
        # stride sets than config 3's?
        for strides in ((4.0, 8.0, 16.0), (4.0, 16.0), (4.0, 8.0), (8.0, 16.0), (2.0, 8.0), (2.0, 32.0), (16.0, 32.0), (8.0,), (2.0,)):
            row = {"strides": strides}
            for name, nw in (("rule", -1), ("one wave", 1), ("four waves", 4)):
                nat.tune_set("pts_nw", nw)
                row[name] = run(strides, full, 2)["shipped"]["us"]
            print(json.dumps(row))
        return
    if "--sweep" in sys.argv:
        # A/B build only (make -C accv-lab_amd/csrc tune; ACCV_HIP_LIB=.../libaccv_hip_tune.so): per-scale mode of the point splat
        none_ = torch.zeros(B, dtype=torch.int32, device=dev)
        for name, knobs in (("default", {"pts_nw": -1, "pts_th": 16}), ("one wave per tile", {"pts_nw": 1, "pts_th": 16}),
                            ("four waves per tile", {"pts_nw": 4, "pts_th": 16}), ("128 x 8 tiles, one wave each", {"pts_nw": -1, "pts_th": 8})):
            for k, v in knobs.items():
                nat.tune_set(k, v)
            row = {"mode": name}
            # same maps in every mode
            chk = [torch.full((B, int(SH / s_), int(SW / s_)), 0.25, device=dev) for s_ in (4.0, 8.0, 16.0)]
            kk = len(chk)
            nat.check(lib.accv_draw_points_multiscale_f32((ctypes.c_void_p * kk)(*[m.data_ptr() for m in chk]),
                                                          (ctypes.c_int * kk)(*[m.size(1) for m in chk]), (ctypes.c_int * kk)(*[m.size(2) for m in chk]),
                                                          (ctypes.c_float * kk)(4.0, 8.0, 16.0), kk, B, samples.data_ptr(), full.data_ptr(), n, 2, 6.0,
                                                          1.0, nat.HM_GROUP_BOXES_GIVEN, work.data_ptr(), ws_bytes, stream), "points")
            torch.cuda.synchronize()
            sig = [float(m.double().sum()) for m in chk] + [int((m != 0.25).sum()) for m in chk]
            row["checksum"] = sig
            for sc, strides in (("all", (4.0, 8.0, 16.0)), ("s4", (4.0,)), ("s8", (8.0,)), ("s16", (16.0,))):
                row[sc] = {"empty": run(strides, none_, 2)["shipped"]["us"], "8 lanes r=2": run(strides, full, 2)["shipped"]["us"],
                           "in-place": run(strides, full, 2, clear=False)["shipped"]["us"]}
            print(json.dumps(row))
        return
    none = torch.zeros(B, dtype=torch.int32, device=dev)
    one = torch.full((B,), Q, dtype=torch.int32, device=dev)
    brief = "--brief" in sys.argv
    for name, strides in (("all", (4.0, 8.0, 16.0)), ("s4", (4.0,)), ("s8", (8.0,)), ("s16", (16.0,))):
        res = {"no lanes (empty-tile floor)": run(strides, none, 2), "1 lane": run(strides, one, 2), "8 lanes r=2": run(strides, full, 2),
               "8 lanes r=0": run(strides, full, 0), "8 lanes r=2 in-place": run(strides, full, 2, clear=False)}
        if brief:     # one line per case: microseconds per build (+ "!" when a build's map differs from the shipped one)
            res = {k: {n: (str(v[n]["us"]) + ("" if v[n].get("same_as_shipped", True) else "!")) for n in libs} for k, v in res.items()}
        print(json.dumps({"scales": name, **res}))


if __name__ == "__main__":
    main()
