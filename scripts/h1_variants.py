#!/usr/bin/env python3
"""In-process A/B of splat-kernel variants on the C1 workload (interleaved rounds, one process —
cdna_hip_programming.md §5.4 rule 24). Prints ms/batch, frames/s and algorithmic GB/s per variant."""
import argparse
import itertools
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import bench_workloads as wl  # noqa: E402
from accvlab import _amd_native as nat  # noqa: E402
from accvlab.draw_heatmap import draw_heatmap_batched  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rule", default="A")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--warm", type=int, default=300, help="untimed launches of a variant before its timed ones")
    ap.add_argument("--knobs", default="",
                    help="knobs of the A/B build (make -C accv-lab_amd/csrc tune; run with "
                         "ACCV_HIP_LIB=accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so): hm_wpg=1|4, hm_rows=8|16, "
                         "hm_nt=0|1|2|4, e.g. 'hm_wpg=1,4;hm_nt=0,4'.  Empty: the shipped dispatch only")
    ap.add_argument("--empty", action="store_true", help="no objects: isolates the store pattern")
    ap.add_argument("--nmin", type=int, default=1, help="minimum objects per frame (128 = densest case)")
    ap.add_argument("--alt-lib", default=None,
                    help="second build of libaccv_hip.so (e.g. the previous commit's) timed beside the shipped one in the "
                         "same process: A/B of CODE changes on one box, since boxes differ by more than most changes")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, W = args.batch, 1080, 1920
    cl, rl = wl.heatmap_objects(B, H, W, args.nmin, 128, args.rule, seed=42)
    cpad, sizes = wl.pad_ragged(cl)
    if args.empty:
        sizes = torch.zeros_like(sizes)
    rpad, _ = wl.pad_ragged(rl)
    c = SimpleNamespace(tensor=cpad.to(dev), sample_sizes=sizes.to(dev))
    r = SimpleNamespace(tensor=rpad.to(dev), sample_sizes=sizes.to(dev))
    hm = torch.zeros((B, H, W), device=dev)
    nbytes = B * H * W * 4
    knobs = []
    for part in [p for p in args.knobs.split(";") if p]:
        k, vals = part.split("=")
        knobs.append([(k, int(v)) for v in vals.split(",")])
    variants = list(itertools.product(*knobs)) if knobs else [()]
    has_knobs = hasattr(nat.ctypes_lib(), "accv_tune_set")
    if knobs and not has_knobs:
        raise SystemExit("--knobs needs the A/B build: make -C accv-lab_amd/csrc tune && ACCV_HIP_LIB=.../libaccv_hip_tune.so")
    lib = nat.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        # warm up with the SAME variant for ~30 ms: the clocks follow the instruction mix of the last tens of
        # milliseconds, so a compute-heavier variant timed right after a store-only one reads up to 10 % slow
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

    alt = None
    if args.alt_lib:
        import ctypes

        alt = ctypes.CDLL(os.path.abspath(args.alt_lib))
        fn = alt.accv_draw_heatmap_batched_f32
        fn.restype, fn.argtypes = nat.SIGNATURES["accv_draw_heatmap_batched_f32"]
        i64 = nat.HM_COUNTS_I64 if c.sample_sizes.dtype == torch.int64 else 0

        def draw_with(handle, clear):
            handle.accv_draw_heatmap_batched_f32(hm.data_ptr(), B, 0, H, W, c.tensor.data_ptr(), r.tensor.data_ptr(),
                                                 c.sample_sizes.data_ptr(), None, r.tensor.shape[1], 6.0, 1.0,
                                                 (nat.HM_CLEAR if clear else 0) | i64, stream)

    res = {}
    for rnd in range(args.rounds):
        t = timed(lambda: lib.accv_fill_f32(hm.data_ptr(), hm.numel(), 0.0, stream))
        res.setdefault(("fill",), []).append(t)
        t = timed(lambda: hm.zero_())
        res.setdefault(("torch.zero_",), []).append(t)
        for var in variants:
            for k, v in var:
                nat.tune_set(k, v)
            for mode in ("clear", "inplace"):
                t = timed(lambda: draw_heatmap_batched(hm, c, r, 6.0, 1.0, clear=(mode == "clear")))
                res.setdefault(var + (mode,), []).append(t)
        if alt is not None:  # both builds through the bare C-ABI, default knobs
            for k, v in (("hm_wpg", 1), ("hm_rows", -1), ("hm_nt", -1)):
                if has_knobs:
                    nat.tune_set(k, v)
            for mode in ("clear", "inplace"):
                for name, handle in (("shipped", lib), ("alt", alt)):
                    t = timed(lambda: draw_with(handle, mode == "clear"))
                    res.setdefault((name + "-lib", mode), []).append(t)
    for key, ts in res.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print(f"{str(key):60s} median {med:8.4f} ms  min {ts[0]:8.4f} ms  {B / med * 1e3:10.0f} frames/s  "
              f"{nbytes / med / 1e6:8.1f} GB/s (write-once bytes)")


if __name__ == "__main__":
    main()
