#!/usr/bin/env python3
"""3000 copies of the 10 000-leaf structure of configs[2] with output recycling on — alternating inline / background mode, new
metadata every step, results kept or dropped: host RSS, device memory and the number of python objects must not grow."""
import os, sys, resource, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]
import torch
import bench_workloads as wl
from accvlab.multi_tensor_copier import start_copy
dev = torch.device("cuda", 0)
tree = wl.meta_tensor_tree(10000, seed=0)
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
res = None
for it in range(3001):
    # alternate: keep / drop, new metadata every step
    for s in tree[:50]: s["meta"]["id"] = 100000 + it
    res = start_copy(tree, dev, use_background_thread=(it % 2 == 0)).get()
    if it % 7 == 0: del res; res = None
    if it % 500 == 0:
        torch.cuda.synchronize(); gc.collect()
        print(it, "rss MB", rss(), "gpu alloc MB", round(torch.cuda.memory_allocated() / 2**20, 1), "reserved MB", round(torch.cuda.memory_reserved() / 2**20, 1), "py objects", len(gc.get_objects()))
ok = all(torch.equal(a.cpu(), b) for a, b in zip(res[3]["gt"], tree[3]["gt"])) and res[3]["meta"]["id"] == 100000 + 3000
print("values ok", ok)
