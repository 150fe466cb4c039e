#!/usr/bin/env python3
"""Does ONE fused clear+draw launch over the 64-frame headline batch cost more than the same batch cut into 2 / 4 / 8 launches
over consecutive plane ranges of the SAME 531 MB map (so the working set does not change)?  bench.py's strong-scaling shards
read 0.787 of the HBM peak for 32 frames against 0.72 for 64 on one box — this separates "smaller launches run better" from
"a 265 MB map partly lives in the 256 MB Infinity Cache".  Bare C-ABI calls, HIP events, interleaved rounds."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402
from accvlab import _amd_native as nat  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, H, W = 64, 1080, 1920
    cl, rl = wl.heatmap_objects(B, H, W, 1, 128, "A", seed=42)
    cpad, sizes = wl.pad_ragged(cl)
    rpad, _ = wl.pad_ragged(rl)
    c, r, n = cpad.to(dev), rpad.to(dev), sizes.to(dev)
    nmax = r.shape[1]
    hm = torch.empty((B, H, W), device=dev)
    hm2 = torch.empty((B, H, W), device=dev)
    lib = nat.lib()
    stream = torch.cuda.current_stream().cuda_stream
    flags = nat.HM_CLEAR | nat.HM_COUNTS_I64

    def draw(buf, lo, hi):
        nat.check(lib.accv_draw_heatmap_batched_f32(buf.data_ptr() + lo * H * W * 4, hi - lo, 0, H, W, c.data_ptr() + lo * nmax * 8,
                                                    r.data_ptr() + lo * nmax * 4, n.data_ptr() + lo * 8, None, nmax, 6.0, 1.0,
                                                    flags, stream), "draw")

    def split(parts, buf=hm):
        step = B // parts
        for k in range(parts):
            draw(buf, k * step, (k + 1) * step)

    def timed(fn, warm=300, iters=200):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    import ctypes
    alts = {}
    for path in [a for a in sys.argv[1:] if a.endswith(".so")]:      # further builds of the library: shard sizes, same process
        h = ctypes.CDLL(os.path.abspath(path))
        h.accv_draw_heatmap_batched_f32.restype, h.accv_draw_heatmap_batched_f32.argtypes = nat.SIGNATURES["accv_draw_heatmap_batched_f32"]
        alts[os.path.basename(path).replace("libaccv_hip_", "").replace(".so", "")] = h
    if "--tile-rows" in sys.argv:      # 128 x 16 tiles (default) against 128 x 32 tiles, per shard size of the strong split
        for frames in (8, 16, 32, 64):
            row = {}
            for name, extra in (("128x16 tiles", nat.HM_TILE_ROWS_8), ("128x32 tiles", nat.HM_TILE_ROWS_16)):
                def one(lo=0, hi=frames, extra=extra):
                    nat.check(lib.accv_draw_heatmap_batched_f32(hm.data_ptr() + lo * H * W * 4, hi - lo, 0, H, W, c.data_ptr() + lo * nmax * 8,
                                                                r.data_ptr() + lo * nmax * 4, n.data_ptr() + lo * 8, None, nmax, 6.0, 1.0,
                                                                flags | extra, stream), "draw")
                per = []
                for k in range(B // frames):
                    per.append(timed(lambda: one(k * frames, (k + 1) * frames), warm=100, iters=200))
                row[name] = {"slowest_us": round(max(per) * 1e3, 2), "fastest_us": round(min(per) * 1e3, 2)}
            print(json.dumps({"frames_per_launch": frames, **row}), flush=True)
        return
    if alts:
        libs = {"shipped": lib, **alts}
        for frames in (8, 16, 32, 64):
            row = {}
            for name, handle in libs.items():
                def one(lo=0, hi=frames, handle=handle):
                    nat.check(handle.accv_draw_heatmap_batched_f32(hm.data_ptr() + lo * H * W * 4, hi - lo, 0, H, W, c.data_ptr() + lo * nmax * 8,
                                                                   r.data_ptr() + lo * nmax * 4, n.data_ptr() + lo * 8, None, nmax, 6.0, 1.0,
                                                                   flags, stream), "draw")
                # every shard of the split, slowest one reported (as bench.py's strong-scaling prediction does)
                per = []
                for k in range(B // frames):
                    per.append(timed(lambda: one(k * frames, (k + 1) * frames), warm=100, iters=200))
                row[name] = {"slowest_us": round(max(per) * 1e3, 2), "fastest_us": round(min(per) * 1e3, 2)}
            print(json.dumps({"frames_per_launch": frames, **row}))
        return
    variants = {
        "1 x 64 frames": lambda: split(1),
        "2 x 32 frames (same 531 MB map)": lambda: split(2),
        "4 x 16 frames (same map)": lambda: split(4),
        "8 x 8 frames (same map)": lambda: split(8),
        "32 frames, one 265 MB map over and over": lambda: draw(hm, 0, 32),
        "32 frames, alternating two 265 MB maps": None,
        "16 frames, one 133 MB map over and over": lambda: draw(hm, 0, 16),
    }
    flip = [0]

    def alternating():
        flip[0] ^= 1
        draw(hm if flip[0] else hm2, 0, 32)

    variants["32 frames, alternating two 265 MB maps"] = alternating
    res = {k: [] for k in variants}
    for _ in range(3):
        for k, fn in variants.items():
            res[k].append(timed(fn))
    ref = None
    for k, v in res.items():
        v.sort()
        med = v[1]
        print(json.dumps({"variant": k, "ms_median": round(med, 5), "ms_min": round(v[0], 5), "all": [round(x, 5) for x in v]}))
    # same values whichever way the batch is cut
    split(1, hm)
    split(4, hm2)
    torch.cuda.synchronize()
    print(json.dumps({"split launches write the same map": bool(torch.equal(hm, hm2))}))


if __name__ == "__main__":
    main()
