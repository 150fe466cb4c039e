#!/usr/bin/env python3
"""splat_small_kernel (ACCV_HM_SMALL_RADII) on integer lane targets of BASELINE config 3, one scale per call: microseconds per
call for radius 0 / 2 / 5, fused clear and in place, through the bare C-ABI; further builds of the library given as `*.so`
arguments are timed in the same process and their maps compared."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import RaggedBatch  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_batched, sample_lanes  # noqa: E402


def gpu_us(fn, n=200):
    for _ in range(30):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 1)


def main():
    dev = torch.device("cuda", 0)
    B, SH, SW, L, P, Q = 32, 2160, 3840, 8, 24, 256
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(B, L, 1, generator=g) * SW
    t_ = torch.linspace(0, 1, P).view(1, 1, P)
    xs = x0 + (torch.rand(B, L, 1, generator=g) - 0.5) * SW * 0.5 * t_ + 60 * torch.sin(6 * t_ + x0)
    ys = SH * (1 - 0.9 * t_).expand(B, L, P)
    samples = sample_lanes(torch.stack([xs, ys], -1).to(dev), Q)           # [B, L*Q, 2] float
    n = torch.full((B,), L * Q, dtype=torch.int64, device=dev)
    libs = {"shipped": nat.ctypes_lib()}
    for path in [a for a in sys.argv[1:] if a.endswith(".so")]:
        h = ctypes.CDLL(os.path.abspath(path))
        h.accv_draw_heatmap_batched_f32.restype, h.accv_draw_heatmap_batched_f32.argtypes = nat.SIGNATURES["accv_draw_heatmap_batched_f32"]
        libs[os.path.basename(path).replace("libaccv_hip_", "").replace(".so", "")] = h
    stream = torch.cuda.current_stream().cuda_stream
    for stride in (4, 8, 16):
        c = (samples / stride).to(torch.int32).contiguous()
        hm = torch.zeros(B, SH // stride, SW // stride, device=dev)
        for r in (0, 2, 5):
            rad = torch.full((B, L * Q), r, dtype=torch.int32, device=dev)
            row, ref = {}, None
            for name, lib in libs.items():
                def call(clear):
                    flags = nat.HM_SMALL_RADII | nat.HM_COUNTS_I64 | (nat.HM_CLEAR if clear else 0)
                    nat.check(lib.accv_draw_heatmap_batched_f32(hm.data_ptr(), B, 0, hm.shape[1], hm.shape[2], c.data_ptr(), rad.data_ptr(),
                                                                n.data_ptr(), None, L * Q, 6.0, 1.0, flags, stream), "draw")
                hm.fill_(0.1)
                t_in = gpu_us(lambda: call(False))
                t_cl = gpu_us(lambda: call(True))
                snap = hm.clone()
                same = True if ref is None else bool(torch.equal(ref.view(torch.int32), snap.view(torch.int32)))
                ref = snap if ref is None else ref
                row[name] = {"clear": t_cl, "in place": t_in, "same": same}
            print(json.dumps({f"stride {stride} r={r}": row}))


if __name__ == "__main__":
    main()
