#!/usr/bin/env python3
"""Does timing every launch with its own start/stop events (accv_draw_heatmap_time_next_launch -> hipExtLaunchKernel) disturb
the stream?  Compares, interleaved, the region time per step with and without the per-launch events, and prints the mean
per-launch kernel duration the events report (what rocprofv3's kernel trace shows) next to the region time per step."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402
from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_batched  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]


def make_events(n):
    out = []
    for _ in range(n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        out.append(e.value)
    return out


def main():
    dev = torch.device("cuda", 0)
    B, H, W, K = 64, 1080, 1920, 500
    cl, rl = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=42)
    c = combine_data(cl, device=dev)
    r = combine_data(rl, device=dev, other_with_same_sample_sizes=c)
    hm = torch.empty(B, H, W, device=dev)
    lib = nat.lib()
    starts, stops = make_events(K), make_events(K)

    def step():
        draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=True)

    def region(instrument):
        for _ in range(300):
            step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(K):
            if instrument:
                lib.accv_draw_heatmap_time_next_launch(starts[i], stops[i])
            step()
        b.record()
        torch.cuda.synchronize()
        per = None
        if instrument:
            ms = ctypes.c_float()
            tot = 0.0
            for i in range(K):
                assert hip.hipEventElapsedTime(ctypes.byref(ms), starts[i], stops[i]) == 0
                tot += ms.value
            per = tot / K
        return a.elapsed_time(b) / K, per

    out = {"plain region ms/step": [], "instrumented region ms/step": [], "per-launch kernel ms": []}
    for _ in range(4):
        out["plain region ms/step"].append(round(region(False)[0], 5))
        reg, per = region(True)
        out["instrumented region ms/step"].append(round(reg, 5))
        out["per-launch kernel ms"].append(round(per, 5))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
