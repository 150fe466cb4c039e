#!/usr/bin/env python3
"""In-process A/B of the PUBLIC dispatch hints of draw_heatmap_batched on the headline workload (configs[1]): tile rows,
write-through stores.  Variants are interleaved round by round and every variant is warmed
with its own launches first (the clocks follow the instruction mix of the last tens of milliseconds)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402
from accvlab import _amd_native as nat  # noqa: E402
from accvlab.batching_helpers import combine_data  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_batched, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rule", default="A")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--warm", type=int, default=300)
    ap.add_argument("--nmin", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, W = args.batch, 1080, 1920
    cl, rl = wl.heatmap_objects(B, H, W, args.nmin, 128, args.rule, seed=42)
    c = combine_data(cl, device=dev)
    r = combine_data(rl, device=dev, other_with_same_sample_sizes=c)
    hm = torch.zeros((B, H, W), device=dev)
    variants = {
        "default": 0,
        "rows8": nat.HM_TILE_ROWS_8,
        "rows16": nat.HM_TILE_ROWS_16,
        "write-through": nat.HM_WRITE_THROUGH,
    }

    def timed(fn):
        for _ in range(args.warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    res, kern = {}, {}
    for _ in range(args.rounds):
        res.setdefault(("torch.zero_", "-"), []).append(timed(lambda: hm.zero_()))
        for name, flags in variants.items():
            ops._FORCED_FLAGS = flags
            for mode in ("clear", "inplace"):
                res.setdefault((name, mode), []).append(timed(lambda: draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=(mode == "clear"))))
                kern[(name, mode)] = nat.last_dispatch()
        ops._FORCED_FLAGS = 0
    nbytes = B * H * W * 4
    for key, ts in res.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print(json.dumps({"variant": key[0], "mode": key[1], "median_ms": round(med, 5), "min_ms": round(ts[0], 5),
                          "frames_per_s": round(B / med * 1e3), "write_once_GBps": round(nbytes / med / 1e6, 1),
                          "kernel": kern.get(key, "")}))


if __name__ == "__main__":
    main()
