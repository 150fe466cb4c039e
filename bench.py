#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): heat-map frames/s + achieved HBM GB/s, 1920x1080 fp32, ragged
N_obj in [1,128] packed by batching_helpers, batch 64 per GPU (configs[1]); frames shard across ranks with
no data-path collective.  Two scaling modes, both built from accvlab.draw_heatmap.sharding:
  weak   (default, `value`): every rank draws its OWN 64-frame batch (seed 42 + rank) — per-GPU work is fixed;
  strong (`--scaling strong`, and `secondary.strong_scaling` of every weak run): the ONE 64-frame batch of seed 42 is cut with
         `shard_range` into 64 / 32 / 16 / 8 frames per GPU at 1 / 2 / 4 / 8 GPUs.

One "step" = one fused clear+draw of the whole batch through the public operator
``accvlab.draw_heatmap.draw_heatmap_batched(..., clear=True)`` -> C-ABI -> one HIP kernel launch, with the
object lists already resident in HBM.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 500 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

The line is self-certifying: `roofline.kernel` is the dispatch string the library reports for the launches that were
timed, `roofline.achieved` comes from HIP events on the launch stream inside this run, `roofline.traffic` is only
quoted from a committed PMC profile of THE SAME kernel instantiation (else null), and the §8(d) work counters, the
rule-B and in-place rates and the CPU baselines (all cores and one core) are measured in this run.
The multi-rank control flow (rank layout, per-rank seeds, barrier-bracketed timed region, MAX over ranks) lives in
accvlab.draw_heatmap.sharding and is exercised on two gloo ranks by tests/test_sharding_gloo.py.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "accv-lab_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

import bench_workloads as wl  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
# v_exp_f32 issues one wave64 instruction per 8 cycles per SIMD (same guide, "vector-instruction ISSUE cost"):
# 256 CUs x 4 SIMDs x 64 lanes / 8 cycles x 2.4 GHz
EXP_PEAK_PER_S = 256 * 4 * 64 / 8 * 2.4e9
H, W = 1080, 1920
TRAFFIC_PROFILES = [os.path.join("profiles", f) for f in ("r03_traffic.json", "r02_traffic.json")]   # newest first
STRONG_TOTAL = 64   # frames of the C4 batch that the strong mode cuts over the ranks


def cpu_baseline(centers_l, radii_l, batch):
    """The CPU oracle (oracle/h1_splat.c, a port of the reference semantics — kind 'port') timed on this host's cores over
    the same batch (bounded sample), all cores (OpenMP over frames) and ONE core (SURVEY §8d asks for both)."""
    import numpy as np

    from oracle import h1 as oracle

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, oracle.max_threads(), batch))
    cpad, sizes = wl.pad_ragged(centers_l)
    rpad, _ = wl.pad_ragged(radii_l)
    hm = np.empty((batch, H, W), dtype=np.float32)
    c, r, s = cpad.numpy(), rpad.numpy(), sizes.numpy()
    oracle.draw_heatmap_batched(hm[:2], c[:2], r[:2], s[:2], clear=True, threads=min(2, cores))  # page-in / warm
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while passes < 16 and dt * cores < 12.0:      # bounded: ~12-24 s of CPU work (cores x wall)
        oracle.draw_heatmap_batched(hm, c, r, s, clear=True, threads=cores)
        passes += 1
        dt = time.perf_counter() - t0
    # one core: a bounded number of leading frames
    n1, dt1 = 0, 0.0
    t0 = time.perf_counter()
    while n1 < batch and dt1 < 4.0:
        k = min(4, batch - n1)
        oracle.draw_heatmap_batched(hm[n1:n1 + k], c[n1:n1 + k], r[n1:n1 + k], s[n1:n1 + k], clear=True, threads=1)
        n1 += k
        dt1 = time.perf_counter() - t0
    return {"value": passes * batch / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{passes} fused clear+draw passes over the full {batch}-frame C1 batch (rule A), "
                      f"{dt:.2f} s wall = {dt * cores:.0f} core-seconds, OpenMP over frames",
            "single_thread": {"value": n1 / dt1, "unit": "frames/s", "cores": 1,
                              "sample": f"first {n1} frames of the same batch, {dt1:.2f} s"}}


def work_counters(centers_l, radii_l, tile_w, tile_h):
    """SURVEY §8(d) work counters from the inputs (numpy, exact): sum of clipped (2r+1)^2 per frame, tiles touched,
    objects-per-tile histogram, and the number of exp evaluations the tile kernel issues for them."""
    import numpy as np

    tx, ty = -(-W // tile_w), -(-H // tile_h)
    area_sum, frames = 0, len(radii_l)
    hits = np.zeros((frames, ty, tx), dtype=np.int32)
    touched_px = 0
    for f, (c, r) in enumerate(zip(centers_l, radii_l)):
        c, r = c.numpy().astype(np.int64), r.numpy().astype(np.int64)
        x0, x1 = np.clip(c[:, 0] - r, 0, W), np.clip(c[:, 0] + r + 1, 0, W)   # the reference's clipped box (cuh:64-67)
        y0, y1 = np.clip(c[:, 1] - r, 0, H), np.clip(c[:, 1] + r + 1, 0, H)
        ok = (x1 > x0) & (y1 > y0) & (r >= 0)
        area_sum += int(((x1 - x0) * (y1 - y0))[ok].sum())
        for a0, a1, b0, b1 in zip(x0[ok] // tile_w, (x1[ok] - 1) // tile_w, y0[ok] // tile_h, (y1[ok] - 1) // tile_h):
            hits[f, b0:b1 + 1, a0:a1 + 1] += 1
    touched = hits > 0
    # pixels of touched tiles (tiles on the bottom edge are clipped to the frame)
    rows = np.minimum(tile_h, H - np.arange(ty) * tile_h)
    cols = np.minimum(tile_w, W - np.arange(tx) * tile_w)
    touched_px = int((touched * rows[None, :, None] * cols[None, None, :]).sum())
    hist = np.bincount(np.minimum(hits.reshape(-1), 16), minlength=17)
    return {"tile": [tile_w, tile_h], "clipped_area_sum_per_frame": area_sum / frames,
            "tiles_total": int(hits.size), "tiles_touched": int(touched.sum()),
            "tile_hits_total": int(hits.sum()), "objects_per_tile_mean": float(hits.mean()),
            "objects_per_tile_max": int(hits.max()),
            "objects_per_tile_histogram_0_to_16plus": hist.tolist(), "touched_pixels": touched_px}


def device_env(dev_index):
    """Clock / power / partition state of the GPU from sysfs (plain file reads: no child process), sampled while the
    queue is full.  Boxes of the pool differ by up to 13 % on this kernel; this records what the box looked like."""
    import glob

    def read(path):
        try:
            with open(path) as fh:
                return fh.read().strip()
        except OSError:
            return None

    def current(path):  # pp_dpm_* files list the levels, the active one is starred
        txt = read(path)
        if not txt:
            return None
        for line in txt.splitlines():
            if line.rstrip().endswith("*"):
                return line.split(":", 1)[-1].replace("*", "").strip()
        return None

    env = {}
    try:
        props = torch.cuda.get_device_properties(dev_index)
        env["name"] = props.name
        base = None
        bus = getattr(props, "pci_bus_id", None)
        if bus is not None:
            cand = glob.glob(f"/sys/bus/pci/devices/{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:"
                             f"{getattr(props, 'pci_device_id', 0):02x}.0")
            base = cand[0] if cand else None
        if base is None:
            cards = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if read(c + "/vendor") == "0x1002"]
            base = cards[dev_index] if dev_index < len(cards) else (cards[0] if cards else None)
        if base:
            for key, f in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
                v = current(f"{base}/{f}")
                if v:
                    env[key] = v
            for key, f in (("compute_partition", "current_compute_partition"),
                           ("memory_partition", "current_memory_partition")):
                v = read(f"{base}/{f}")
                if v:
                    env[key] = v
            for hw in glob.glob(f"{base}/hwmon/hwmon*"):
                for key, f, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6),
                                      ("power_cap_w", "power1_cap", 1e-6)):
                    v = read(f"{hw}/{f}")
                    if v and v.isdigit() and key not in env:
                        env[key] = round(int(v) * scale, 1)
    except Exception as e:  # noqa: BLE001 - diagnostics only
        env["error"] = str(e)
    return env


def committed_traffic(kernel: str, frames: int):
    """HBM bytes per launch from the newest committed PMC profile that was taken on the SAME kernel instantiation as the one
    timed now (the file names it); otherwise (None, reason)."""
    want = kernel.split(" grid")[0]
    reasons = []
    for rel in TRAFFIC_PROFILES:
        try:
            rec = json.load(open(os.path.join(ROOT, rel)))
        except Exception:  # noqa: BLE001
            reasons.append(f"{rel} not present")
            continue
        if rec.get("kernel") != want:
            reasons.append(f"{rel} was taken on {rec.get('kernel')!r}, this run dispatched {want!r}")
            continue
        if int(rec.get("frames", 64)) != frames:
            reasons.append(f"{rel} was taken on {rec.get('frames', 64)} frames per launch, this run draws {frames}")
            continue
        return rec.get("hbm_bytes_per_launch"), (f"{rel} (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes, commit "
                                                 f"{rec.get('commit')}); not measured in this run")
    return None, "; ".join(reasons)


def per_launch_kernel_ms(nat, step, steps, sync):
    """Mean duration of the splat kernel over `steps` back-to-back launches, each bracketed by its own pair of HIP events that
    the library attaches to the KERNEL (start = kernel begins, stop = kernel done).  (None, reason) if the HIP runtime cannot
    be reached through ctypes."""
    import ctypes

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    except OSError as exc:       # pragma: no cover
        return None, f"libamdhip64 not loadable: {exc}"
    events = []
    for _ in range(2 * steps):
        e = ctypes.c_void_p()
        if hip.hipEventCreate(ctypes.byref(e)) != 0:
            return None, "hipEventCreate failed"
        events.append(e.value)
    lib = nat.lib()
    for _ in range(150):         # the closing barrier of the timed region drained the queue: refill it before timing kernels
        step()
    for i in range(steps):
        nat.check(lib.accv_draw_heatmap_time_next_launch(events[2 * i], events[2 * i + 1]), "time_next_launch")
        step()
    sync()
    total, ms = 0.0, ctypes.c_float()
    for i in range(steps):
        if hip.hipEventElapsedTime(ctypes.byref(ms), events[2 * i], events[2 * i + 1]) != 0:
            return None, "hipEventElapsedTime failed"
        total += ms.value
    for e in events:
        hip.hipEventDestroy(e)
    return total / steps, ("mean over %d launches behind the timed region, each with start/stop HIP events on the kernel itself "
                           "(hipExtLaunchKernel); the events space the launches ~4 us apart and a spaced launch runs faster "
                           "than a back-to-back one — not the roofline figure" % steps)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU")
    ap.add_argument("--rule", default="A", choices=["A", "B"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="which mode `value` reports; the other one is carried in `secondary`")
    ap.add_argument("--no-configs", action="store_true", help="skip configs[0] / [2] / [3] (secondary.configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the rule-B / in-place / zero-fill side measurements")
    args = ap.parse_args()

    from accvlab.draw_heatmap import sharding

    rank, local_rank, world = sharding.rank_layout()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # one process per GPU; ACCV_BENCH_BACKEND=gloo lets several ranks share a GPU to REHEARSE the multi-rank control flow
    # on a one-GPU box (the real runs use RCCL = backend "nccl")
    backend = os.environ.get("ACCV_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = sharding.init_process_group(world, backend, dev)   # only the barrier and the max-over-ranks use it

    from accvlab import _amd_native as nat
    from accvlab.batching_helpers import combine_data
    from accvlab.draw_heatmap import draw_heatmap_batched

    B = args.batch
    seed = sharding.rank_seed(42, rank)
    centers_l, radii_l = wl.heatmap_objects(B, H, W, 1, 128, args.rule, seed=seed)
    centers = combine_data(centers_l, device=dev)                       # RaggedBatch i32 [B, Nmax, 2]
    radii = combine_data(radii_l, device=dev, other_with_same_sample_sizes=centers)
    n_objects = int(sum(int(r.shape[0]) for r in radii_l))
    hm = torch.empty((B, H, W), dtype=torch.float32, device=dev)

    launches = [0]    # fused clear+draw launches issued so far (lets the rocprofv3 summary find the timed region in a trace)

    def step():
        launches[0] += 1
        draw_heatmap_batched(hm, centers, radii, 6.0, 1.0, clear=True)

    sync = torch.cuda.synchronize
    for _ in range(args.warmup):
        step()
    # the box's clock / power state, sampled with work in the queue.  The sysfs reads take ~0.1 s during which the queue runs
    # dry, so this comes BEFORE the pre-warm, never between the pre-warm and the timed region
    # (every rank does it, so that all ranks reach the pre-warm and the opening barrier of the timed region together; rank 0
    # reports its sample)
    for _ in range(300):
        step()
    env = device_env(dev_index)
    # extra untimed pre-warm: the chip needs ~15 ms of back-to-back launches after an idle/sync before kernel times
    # settle (profiles/r01_rocprof: 109 -> 128 -> 97 us); keep the queue full for >= 100 ms right up to the timed region
    sync()
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.1:
        for _ in range(50):
            step()
        sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_first = torch.cuda.Event(enable_timing=True)
    trace_index = {"timed_region_launches": args.steps}

    def timed_step_region():
        # HIP events on the CURRENT stream = the stream the kernel is launched on; recorded inside the barrier bracket
        trace_index["timed_region_first_launch"] = launches[0]
        e0.record()
        for i in range(args.steps):
            step()
            if i == 0:
                e_first.record()
        e1.record()

    # exactly K steps between barrier + synchronize brackets; wall clock of this rank, MAX over ranks below
    wall_ms = sharding.timed_steps(timed_step_region, 1, dist=dist, sync=sync) / args.steps
    # Kernel duration from the events of the timed region.  The opening bracket leaves the GPU idle, and the first launch only
    # starts once the host has pushed it through (tens of microseconds that belong to no kernel: with K = 20 they inflate
    # (e1 - e0) / K by 3-5 %, profiles/r02_k20_trace.log).  From the end of launch 1 to the end of launch K the stream runs
    # back to back, so launches 2..K give the duration of a launch in the timed region; (e1 - e0) / K is reported beside it.
    region_ms = e0.elapsed_time(e1) / args.steps
    kern_ms = e_first.elapsed_time(e1) / (args.steps - 1) if args.steps > 1 else region_ms
    kernel = nat.last_dispatch()
    # for information: the same kernel timed launch by launch with start/stop events on the kernel itself (hipExtLaunchKernel
    # through accv_draw_heatmap_time_next_launch), K further launches behind the timed region.  Those events cost ~4 us of
    # dispatch each, so the launches are SPACED — and a spaced launch runs 3-4 % faster than a back-to-back one (rocprofv3 shows
    # the same split: profiles/r02_rocprof).  The roofline entry stays on the back-to-back figure, i.e. on what a step costs.
    trace_index["spaced_first_launch"] = launches[0] + 150     # per_launch_kernel_ms refills the queue with 150 launches first
    trace_index["spaced_launches"] = args.steps
    isolated_ms, isolated_note = per_launch_kernel_ms(nat, step, args.steps, sync)
    red_dev = dev if backend == "nccl" else None
    wall_ms = sharding.max_over_ranks(wall_ms, device=red_dev)
    kern_ms = sharding.max_over_ranks(kern_ms, device=red_dev)
    region_ms = sharding.max_over_ranks(region_ms, device=red_dev)

    # ---- strong scaling (every rank, same collectives in the same order): the 64 frames of seed 42, cut over the ranks
    def strong_inputs(begin, end, all_lists=[]):
        if not all_lists:
            all_lists.extend(wl.heatmap_objects(STRONG_TOTAL, H, W, 1, 128, args.rule, seed=42))
        cl, rl = all_lists[0][begin:end], all_lists[1][begin:end]
        c = combine_data(cl, device=dev)
        r = combine_data(rl, device=dev, other_with_same_sample_sizes=c)
        return c, r, int(sum(int(x.shape[0]) for x in rl))

    strong_buf = []

    def strong_step(begin, end):
        c, r, _ = strong_inputs(begin, end)
        if hm.shape[0] < end - begin and (not strong_buf or strong_buf[0].shape[0] < end - begin):
            strong_buf[:] = [torch.empty((end - begin, H, W), dtype=torch.float32, device=dev)]   # --batch below the shard size
        view = (hm if hm.shape[0] >= end - begin else strong_buf[0])[: end - begin]
        return lambda: draw_heatmap_batched(view, c, r, 6.0, 1.0, clear=True)

    strong = None
    if world > 1 or args.scaling == "strong":
        strong = sharding.strong_scaling_run(strong_step, STRONG_TOTAL, rank, world, args.steps, args.warmup + 300,
                                             dist=dist, sync=sync, device=red_dev)
        strong["kernel"] = nat.last_dispatch()

    # secondary, rank 0 only: rule B, the reference's exact in-place semantics, and the streaming-write ceiling
    extra = {}
    if rank == 0 and not args.no_secondary:
        def timed(fn, iters=50, warm=20):
            for _ in range(warm):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sync()
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            sync()
            return a.elapsed_time(b) / iters

        frame_bytes = H * W * 4
        ms_inplace = timed(lambda: draw_heatmap_batched(hm, centers, radii, 6.0, 1.0))
        k_inplace = nat.last_dispatch()
        ms_zero = timed(lambda: hm.zero_())
        wc_in = work_counters(centers_l, radii_l, 128, 16 if "R=8" in k_inplace else 32)
        in_bytes = 8 * wc_in["touched_pixels"] + 12 * n_objects + 4 * B       # read + write of touched tiles only
        extra = {
            "inplace": {"frames_per_s": B / ms_inplace * 1e3, "ms": ms_inplace, "kernel": k_inplace,
                        "algorithmic_bytes": in_bytes, "achieved_GBps": in_bytes / ms_inplace / 1e6,
                        "frac": in_bytes / ms_inplace / 1e6 / HBM_PEAK_GBPS,
                        "note": "reference semantics (max into the existing map): 8 B per pixel of every touched tile, "
                                "untouched tiles cost nothing (SURVEY 8d)"},
            "zero_then_inplace_frames_per_s": B / (ms_inplace + ms_zero) * 1e3,
            "torch_zero_fill_GBps": hm.numel() * 4 / ms_zero / 1e6, "torch_zero_fill_ms": ms_zero,
        }
        other = "B" if args.rule == "A" else "A"
        cb_l, rb_l = wl.heatmap_objects(B, H, W, 1, 128, other, seed=seed)
        cb = combine_data(cb_l, device=dev)
        rbb = combine_data(rb_l, device=dev, other_with_same_sample_sizes=cb)
        nb = int(sum(int(r.shape[0]) for r in rb_l))
        ms_b = timed(lambda: draw_heatmap_batched(hm, cb, rbb, 6.0, 1.0, clear=True))
        k_b = nat.last_dispatch()
        ms_b_in = timed(lambda: draw_heatmap_batched(hm, cb, rbb, 6.0, 1.0))
        wc_b = work_counters(cb_l, rb_l, 128, 16)
        b_bytes = B * frame_bytes + 12 * nb + 4 * B
        b_in_bytes = 8 * wc_b["touched_pixels"] + 12 * nb + 4 * B
        extra[f"rule_{other}"] = {
            "objects": nb, "clear": {"frames_per_s": B / ms_b * 1e3, "ms": ms_b, "kernel": k_b, "algorithmic_bytes": b_bytes,
                                     "achieved_GBps": b_bytes / ms_b / 1e6, "frac": b_bytes / ms_b / 1e6 / HBM_PEAK_GBPS},
            "inplace": {"frames_per_s": B / ms_b_in * 1e3, "ms": ms_b_in, "algorithmic_bytes": b_in_bytes,
                        "achieved_GBps": b_in_bytes / ms_b_in / 1e6},
            "clipped_area_sum_per_frame": wc_b["clipped_area_sum_per_frame"], "tiles_touched": wc_b["tiles_touched"]}

    if rank == 0 and world == 1 and not args.no_secondary:
        # what a shard of the strong split costs on ONE GPU (all shards of each split, slowest one counts): predicts the
        # strong curve the driver measures — an 8-frame launch is 66 MB of map, i.e. store stream + launch boundary
        def shard_ms(begin, end, iters=200, warm=100):
            fn = strong_step(begin, end)
            for _ in range(warm):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            sync()
            return a.elapsed_time(b) / iters

        pred = {}
        for n in (1, 2, 4, 8):
            per = [shard_ms(*sharding.shard_range(STRONG_TOTAL, r, n)) for r in range(n)]
            frames = sharding.shard_range(STRONG_TOTAL, 0, n)[1]
            pred[str(n)] = {"frames_per_gpu": frames, "ms_slowest_shard": max(per), "ms_fastest_shard": min(per),
                            "predicted_frames_per_s": STRONG_TOTAL / max(per) * 1e3,
                            "frac_of_hbm_peak_slowest_shard": frames * H * W * 4 / max(per) / 1e6 / HBM_PEAK_GBPS}
        base = pred["1"]["predicted_frames_per_s"]
        for n in pred:
            pred[n]["predicted_speedup"] = pred[n]["predicted_frames_per_s"] / base
        extra["strong_scaling_prediction_from_one_gpu"] = {
            "splits": pred, "note": "each shard of the seed-42 64-frame batch drawn back to back on this GPU (HIP events, 200 "
                                    "launches); N ranks run their shards concurrently, so the job takes the slowest shard's time"}
    if rank == 0 and world == 1 and not args.no_secondary and not args.no_configs:
        import bench_configs

        t_cfg = time.perf_counter()
        extra["configs"] = bench_configs.run()
        extra["configs_wall_s"] = time.perf_counter() - t_cfg

    if rank != 0:
        if dist is not None:
            dist.barrier()          # wait for rank 0's secondary measurements, then leave together
            dist.destroy_process_group()
        return

    alg_bytes = B * H * W * 4 + 12 * n_objects + 4 * B  # H*W*4 + 12*N_i + 4 per frame
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, traffic_source = committed_traffic(kernel, B)
    tile_h = 32 if "R=16" in kernel else 16
    wc = work_counters(centers_l, radii_l, 128, tile_h)
    # exp evaluations of the tile kernel: per (tile, hit) one row-factor table entry per tile row + 4 column factors per lane
    exps = wc["tile_hits_total"] * (tile_h + 256)
    wc["exp_evaluations_per_launch"] = exps
    wc["transcendental_ceiling"] = {"peak_exp_per_s": EXP_PEAK_PER_S, "min_ms_per_launch": exps / EXP_PEAK_PER_S * 1e3,
                                    "fraction_of_kernel_time": exps / EXP_PEAK_PER_S * 1e3 / kern_ms}
    weak = {"frames_per_s": sharding.job_throughput(B, world, wall_ms), "ms_per_step": wall_ms, "frames_per_rank": B,
            "total_frames": B * world}
    if strong is not None:
        extra["strong_scaling"] = strong
    extra["weak_scaling"] = weak
    use_strong = args.scaling == "strong"
    out = {
        "metric": "heatmap frames/sec (1920x1080 fp32, ragged N_obj in [1,128], fused clear+draw)",
        "value": strong["frames_per_s"] if use_strong else weak["frames_per_s"],
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": strong["ms_per_step"] if use_strong else wall_ms,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": ({"workload": f"configs[4] strong split of configs[1]: ONE {STRONG_TOTAL}-frame batch (seed 42) of draw_heatmap_batched "
                                f"1920x1080, ragged N_obj in [1,128], radius rule {args.rule}, cut contiguously over {world} GPU(s) "
                                f"({strong['frames_per_rank']} frames per rank), fp32",
                    "frames_per_gpu": strong["frames_per_rank"], "parallelism": f"frame-sharded x{world} (strong)", "seed": 42}
                   if use_strong else
                   {"workload": f"configs[1]: draw_heatmap_batched 1920x1080, batch {B}/GPU, ragged N_obj in [1,128] "
                                f"via batching_helpers.combine_data, radius rule {args.rule}, factor 6, k 1, fp32",
                    "frames_per_gpu": B, "objects_per_gpu": n_objects, "parallelism": f"frame-sharded x{world}",
                    "seed": seed}),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     # the same bytes over (e1 - e0) / K of the timed region and over the barrier-bracketed wall time of `value`
                     "frac_region": alg_bytes / (region_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_wall": alg_bytes / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "frac_definition": "frac = back-to-back launches (kernel_ms); frac_region = (e1 - e0) / K incl. the start-up "
                                        "of launch 1 after the opening barrier; frac_wall = host wall clock of the weak timed region",
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": kernel, "kernel_ms": kern_ms, "algorithmic_bytes": alg_bytes,
                     "kernel_ms_source": "HIP events on the launch stream inside the timed region: end of launch 1 -> end of "
                                         "launch K, divided by K - 1 (back-to-back launches)",
                     "timed_region_event_ms_per_step": region_ms,
                     "kernel_ms_spaced_launches": isolated_ms, "kernel_ms_spaced_launches_note": isolated_note,
                     "trace_index": trace_index},
        "work": wc,
        "secondary": extra,
        "device": env or {},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(centers_l, radii_l, B)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
