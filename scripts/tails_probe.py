#!/usr/bin/env python3
"""Kernel-level A/B of the small-problem tails through the bare C-ABI, shipped build against other builds
of the library in the same process (`tails_probe.py a.so b.so`): segmented mask -> indices on few wide rows, and the polyline
sampler on one / few long polylines.  Outputs and workspaces are allocated once, so the figures are launch + kernel time of
back-to-back calls (HIP events), not the python operators' host time."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab import _amd_native as nat  # noqa: E402


def gpu_us(fn, n=300, warm=50):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 2)


def main():
    dev = torch.device("cuda", 0)
    libs = {"shipped": nat.ctypes_lib()}
    for path in [a for a in sys.argv[1:] if a.endswith(".so")]:
        h = ctypes.CDLL(os.path.abspath(path))
        for name in ("accv_ragged_mask_to_indices_ws", "accv_ragged_mask_to_indices_workspace_bytes", "accv_polyline_sample",
                     "accv_polyline_scratch_bytes"):
            getattr(h, name).restype, getattr(h, name).argtypes = nat.SIGNATURES[name]
        libs[os.path.basename(path).replace("libaccv_hip_", "").replace(".so", "")] = h
    if "--alt-first" in sys.argv:      # measure the other builds before the shipped one (is a hiccup tied to the position in a row?)
        libs = dict(reversed(list(libs.items())))
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    for b, w in ((8, 65536), (2, 131072), (2, 262144), (64, 65536), (1, 8192)):
        mask = (torch.rand(b, w, generator=g) < 0.3).to(dev)
        idx = torch.empty(b, w, dtype=torch.int64, device=dev)
        sizes = torch.empty(b, dtype=torch.int64, device=dev)
        row, ref = {}, None
        for name, lib in libs.items():
            nb = lib.accv_ragged_mask_to_indices_workspace_bytes(b, w)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            fn = lambda: nat.check(lib.accv_ragged_mask_to_indices_ws(mask.data_ptr(), None, 0, b, w, idx.data_ptr(), sizes.data_ptr(),
                                                                        ws.data_ptr(), nb, stream), "m2i")
            idx.fill_(-7)
            row[name] = gpu_us(fn)
            snap = (idx.clone(), sizes.clone())
            if ref is None:
                ref = snap
            elif not (torch.equal(ref[0], snap[0]) and torch.equal(ref[1], snap[1])):
                row[name] = str(row[name]) + "!"
        print(json.dumps({"mask_to_indices": f"{b} x {w}", "us": row}))
    knobs = [{}]
    if "--sweep" in sys.argv:     # A/B build only (ACCV_HIP_LIB=.../libaccv_hip_tune.so)
        knobs = [{"poly_wide": 2048, "poly_spread": 0}, {"poly_wide": 2048, "poly_spread": 512}, {"poly_wide": 1024, "poly_spread": 0},
                 {"poly_wide": 1024, "poly_spread": 512}, {"poly_wide": 4096, "poly_spread": 512}]
    for kn in knobs:
      for k_, v_ in kn.items():
        nat.tune_set(k_, v_)
      if kn:
        print(json.dumps({"knobs": kn}))
      for batch, npts, nq in ((256, 24, 256), (64, 100, 100), (1, 100, 100), (64, 500, 500), (1, 1000, 1000), (1, 2000, 2000), (1, 5000, 1), (1, 5000, 5000), (64, 2000, 2000), (64, 5000, 1)):
          pts = torch.rand(batch, npts, 2, generator=g).cumsum(1).to(dev)
          dist = torch.rand(batch, nq, generator=g).sort(1).values.to(dev)
          out = torch.empty(batch, nq, 2, device=dev)
          row, ref = {}, None
          for name, lib in libs.items():
              nb = lib.accv_polyline_scratch_bytes(batch, npts, 0)
              scr = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
              fn = lambda: nat.check(lib.accv_polyline_sample(pts.data_ptr(), dist.data_ptr(), None, None, out.data_ptr(), None, batch, npts,
                                                                nq, 2, 0, 0, 1, scr.data_ptr(), nb, stream), "poly")
              out.fill_(-7.0)
              row[name] = gpu_us(fn)
              snap = out.clone()
              if ref is None:
                  ref = snap
              elif not torch.equal(ref.view(torch.int32), snap.view(torch.int32)):
                  row[name] = str(row[name]) + "!"
          print(json.dumps({"polyline sample": f"batch {batch}, {npts} points x {nq} distances", "us": row}))


if __name__ == "__main__":
    main()
