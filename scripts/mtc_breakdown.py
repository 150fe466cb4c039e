#!/usr/bin/env python3
"""Where does a start_copy(...).get() of N small CPU tensors spend its time?  Wraps the phases of
accvlab.multi_tensor_copier (inline mode) with wall-clock timers.  GPU needed."""
import argparse
import json
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402
from accvlab import _amd_native as nat  # noqa: E402
from accvlab.multi_tensor_copier import copier  # noqa: E402

acc = defaultdict(float)


def wrap(owner, name, label=None):
    fn = getattr(owner, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0

    setattr(owner, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tensors", type=int, default=10_000)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--background", action="store_true", help="run the orchestration on the pool thread (the default mode)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tree = wl.meta_tensor_tree(a.tensors, seed=0)

    class LibProxy:  # times the C-ABI calls
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            fn = getattr(self._lib, name)
            if not name.startswith("accv_mtc") and not name.startswith("accv_pinned"):
                return fn

            def timed(*args):
                t0 = time.perf_counter()
                try:
                    return fn(*args)
                finally:
                    acc["C-ABI " + name] += time.perf_counter() - t0

            return timed

    real_lib = nat.lib()
    proxy = LibProxy(real_lib)
    nat_lib_orig = nat.lib
    copier._nat.lib = lambda: proxy
    wrap(copier, "_make_leaf_set", "tree build (walk)")
    wrap(copier, "_run", "_run total")
    wrap(copier, "_plan", "plan (numpy prep + C-ABI)")
    wrap(copier, "_prepare_packed_h2d", "prepare packed (plan + chunks + staging blocks + ordering)")
    wrap(copier, "_start_native_h2d", "native background submit (prepare + async stage)")
    wrap(copier, "_rebuild_without_gc", "rebuild (gc paused)")
    tree_cls = copier._host.Tree if copier._host is not None else copier._PyLeafSet
    for m in ("classify", "make_packed_views", "rebuild"):
        wrap(tree_cls, m, f"tree.{m}")
    for _ in range(5):
        copier.start_copy(tree, dev, use_background_thread=a.background).get()
    torch.cuda.synchronize()
    acc.clear()
    t_total = 0.0
    t_get = 0.0
    t_free = 0.0
    for _ in range(a.iters):
        t0 = time.perf_counter()
        h = copier.start_copy(tree, dev, use_background_thread=a.background)
        t1 = time.perf_counter()
        res = h.get()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        del res, h                       # a timing loop that discards its result pays for freeing 10 k tensors each time
        t3 = time.perf_counter()
        t_total += t2 - t0
        t_get += t2 - t1
        t_free += t3 - t2
    acc["free the previous result (del)"] = t_free
    out = {k: round(v / a.iters * 1e3, 4) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}
    out["start_copy+get total ms"] = round(t_total / a.iters * 1e3, 4)
    out["get() (wait + rebuild) ms"] = round(t_get / a.iters * 1e3, 4)
    copier._nat.lib = nat_lib_orig
    print(json.dumps({"tensors": a.tensors, "background": a.background, **out}, indent=1))


if __name__ == "__main__":
    main()
