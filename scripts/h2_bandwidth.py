#!/usr/bin/env python3
"""Ragged byte movers at sizes where they are bandwidth- instead of launch-bound: achieved GB/s (algorithmic bytes =
row read + row write per valid index, + index bytes) against torch's own gather / index_copy on the same data.
At the sizes of the training path these ops are launch bound; this shows what the kernels do when fed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "accv-lab_amd")]

import torch  # noqa: E402

from accvlab.batching_helpers import batched_indexing_access_cuda as ext  # noqa: E402


def timeit(fn, warm=20, iters=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    out_lines = []
    for (b, n, k, row) in ((64, 4096, 4096, 1024), (64, 16384, 8192, 256), (256, 900, 300, 1024), (8, 900, 100, 1024)):
        d = row // 4
        data = torch.randn(b, n, d, device=dev)
        idx = torch.stack([torch.randperm(n, generator=g)[:k] for _ in range(b)]).to(dev)
        counts = torch.randint(k // 2, k + 1, (b,), generator=g).to(dev)
        valid = int(counts.sum())
        out = torch.zeros(b, k, d, device=dev)
        t_g = timeit(lambda: ext.gather_rows(data, idx, counts, k, out))
        dst = torch.zeros(b, n, d, device=dev)
        src = torch.randn(b, k, d, device=dev)
        t_s = timeit(lambda: ext.scatter_rows(src, idx, counts, k, dst))
        # torch formulations on the same (dense, all-valid) problem
        idx_e = idx.unsqueeze(-1).expand(b, k, d)
        t_tg = timeit(lambda: torch.gather(data, 1, idx_e, out=out))
        t_ts = timeit(lambda: dst.scatter_(1, idx_e, src))
        flat = torch.randn(valid, d, device=dev)
        offs = torch.cumsum(counts, 0) - counts
        t_p = timeit(lambda: ext.pack_rows(flat, offs, counts, k))
        bytes_g = valid * (2 * row + 8)
        bytes_dense = b * k * (2 * row + 8)
        bytes_p = valid * row + b * k * row
        out_lines.append({
            "shape": {"batch": b, "rows": n, "indices": k, "row_bytes": row, "valid": valid},
            "ragged_gather_GBps": bytes_g / t_g / 1e9, "ragged_scatter_GBps": bytes_g / t_s / 1e9,
            "torch_gather_GBps": bytes_dense / t_tg / 1e9, "torch_scatter_GBps": bytes_dense / t_ts / 1e9,
            "pack_rows_GBps": bytes_p / t_p / 1e9,
            "us": {"gather": t_g * 1e6, "scatter": t_s * 1e6, "torch_gather": t_tg * 1e6, "torch_scatter": t_ts * 1e6,
                   "pack": t_p * 1e6}})
    # the remaining kernels at one large shape: pad fill (write-only), mask -> indices (ballot compaction), accumulate
    from accvlab.batching_helpers import RaggedBatch

    b, n, row = 64, 8192, 256
    d = row // 4
    data = torch.randn(b, n, d, device=dev)
    counts = torch.randint(n // 4, n + 1, (b,), generator=g).to(dev)
    rb = RaggedBatch(data, sample_sizes=counts)
    t_pad = timeit(lambda: rb.set_padded_to(0.0))
    pad_bytes = int((n - counts).sum()) * row
    mask = (torch.rand(b, n * 8, generator=g) < 0.3).to(dev)
    t_m2i = timeit(lambda: ext.mask_to_indices(mask))
    k = 4096
    idx = torch.randint(0, n, (b, k), generator=g).to(dev)                     # duplicates -> real atomics
    kc = torch.full((b,), k, device=dev)
    src = torch.randn(b, k, d, device=dev)
    t_acc = timeit(lambda: ext.backward_new_tensor(src, idx, kc, n, 0.0, True))
    acc_bytes = b * k * 2 * row + b * n * row                                 # read rows + atomic RMW + zero-filled result
    print(json.dumps({"shape": {"batch": b, "rows": n, "row_bytes": row},
                      "pad_fill_GBps": pad_bytes / t_pad / 1e9, "pad_fill_us": t_pad * 1e6,
                      "mask_to_indices_GBps_of_mask_plus_index_bytes": (mask.numel() + mask.numel() * 8) / t_m2i / 1e9,
                      "mask_to_indices_us": t_m2i * 1e6,
                      "accumulate_scatter_GBps": acc_bytes / t_acc / 1e9, "accumulate_us": t_acc * 1e6}))
    # few, very wide rows (dense anchor masks of a small batch): segmented two-pass kernels
    for (b2, w2) in ((8, 65536), (2, 262144), (64, 65536)):
        mask2 = (torch.rand(b2, w2, generator=g) < 0.3).to(dev)
        t2 = timeit(lambda: ext.mask_to_indices(mask2))
        print(json.dumps({"mask_to_indices": {"batch": b2, "width": w2}, "us": t2 * 1e6,
                          "GBps_of_mask_plus_index_bytes": (mask2.numel() * 9) / t2 / 1e9}))
    # pad fill at a size where the launch ramp no longer matters (the 64 x 8192 case above writes only ~50 MB)
    b3, n3 = 512, 8192
    data3 = torch.randn(b3, n3, d, device=dev)
    counts3 = torch.randint(n3 // 4, n3 + 1, (b3,), generator=g).to(dev)
    rb3 = RaggedBatch(data3, sample_sizes=counts3)
    t3 = timeit(lambda: rb3.set_padded_to(0.0))
    pad3 = int((n3 - counts3).sum()) * row
    print(json.dumps({"pad_fill_large": {"batch": b3, "rows": n3, "row_bytes": row, "padded_bytes": pad3}, "us": t3 * 1e6,
                      "GBps_of_padded_bytes": pad3 / t3 / 1e9}))
    for line in out_lines:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
