#!/usr/bin/env python3
"""Reduce the three rocprofv3 passes of scripts/collect_profiles.sh to one JSON: average duration of the splat kernel
(all launches and the last 500 = bench.py's timed region), WRITE_SIZE / FETCH_SIZE per launch with the gfx950
corrections of MI355X_MICROARCH.md (FETCH_SIZE doubled), next to bench.py's own HIP-event figure from the same run."""
import csv
import glob
import json
import os
import sys


def rows(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        return []
    with open(files[0]) as f:
        return list(csv.DictReader(f))


def main(out):
    res = {}
    trace = rows(os.path.join(out, "trace", "**", "*kernel_trace.csv"))
    # launches of the headline instantiation over the headline batch only: bench.py also draws shards of 8 / 16 / 32 frames
    # (strong-scaling prediction) and the rule-B batch with the same kernel
    frames = 64
    try:
        with open(os.path.join(out, "bench_under_rocprof.json")) as fh:
            frames = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])["config"]["frames_per_gpu"]
    except Exception:  # noqa: BLE001
        pass
    res["frames_per_launch"] = frames
    trace = sorted(trace, key=lambda r: int(r["Start_Timestamp"]))
    clear_all = [r for r in trace if "splat_kernel<" in r["Kernel_Name"] and "true" in r["Kernel_Name"].split("splat_kernel")[1][:20]]
    dur_all = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in clear_all]
    dur = [d for r, d in zip(clear_all, dur_all) if int(r.get("Grid_Size_Z", frames)) == frames]
    if dur:
        res["clear_kernel"] = {"name": next(r["Kernel_Name"] for r in trace if "splat_kernel" in r["Kernel_Name"]),
                               "launches": len(dur), "avg_us_all": sum(dur) / len(dur) / 1e3}
        # bench.py says which of its fused clear+draw launches were the timed region (back to back) and which were the
        # launches timed one by one (spaced by their events): roofline.trace_index of the line it printed under the profiler
        try:
            with open(os.path.join(out, "bench_under_rocprof.json")) as fh:
                idx = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])["roofline"]["trace_index"]
            # (the indices count every fused clear+draw launch of the run in issue order: index into the unfiltered list)
            a, n = idx["timed_region_first_launch"], idx["timed_region_launches"]
            res["clear_kernel"]["avg_us_timed_region_back_to_back"] = sum(dur_all[a:a + n]) / n / 1e3
            b, m = idx["spaced_first_launch"], idx["spaced_launches"]
            res["clear_kernel"]["avg_us_spaced_launches"] = sum(dur_all[b:b + m]) / m / 1e3
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            res["clear_kernel"]["avg_us_last500"] = sum(dur[-500:]) / len(dur[-500:]) / 1e3
    stats = rows(os.path.join(out, "trace", "**", "*kernel_stats.csv"))
    res["kernel_stats_top"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")}
                               for r in stats[:6]]
    for key, sub, counter in (("write", "write", "WRITE_SIZE"), ("fetch", "fetch", "FETCH_SIZE")):
        cr = rows(os.path.join(out, sub, "**", "*counter_collection.csv"))
        per = {}
        for r in cr:
            if r.get("Counter_Name") != counter or "splat_kernel<" not in r["Kernel_Name"]:
                continue
            # Grid_Size = work-items of the launch: tiles_x * tiles_y * frames workgroups of 64 lanes (1920 x 1080: 15 x 68 tiles)
            if int(float(r.get("Grid_Size", 0))) != 15 * 68 * frames * 64:
                continue
            mode = "clear" if ", true," in r["Kernel_Name"] else "inplace"
            per.setdefault(mode, []).append(float(r["Counter_Value"]))
        res[counter + "_KB_per_launch"] = {m: sum(v) / len(v) for m, v in per.items()}
    w = res.get("WRITE_SIZE_KB_per_launch", {}).get("clear")
    f = res.get("FETCH_SIZE_KB_per_launch", {}).get("clear")
    if w is not None and f is not None:
        res["hbm_bytes_per_launch_clear"] = int(w * 1024 + 2 * f * 1024)   # FETCH_SIZE doubled on gfx950
    wi = res.get("WRITE_SIZE_KB_per_launch", {}).get("inplace")
    fi = res.get("FETCH_SIZE_KB_per_launch", {}).get("inplace")
    if wi is not None and fi is not None:
        res["hbm_bytes_per_launch_inplace"] = int(wi * 1024 + 2 * fi * 1024)
    # traffic record for bench.py (roofline.traffic): names the kernel instantiation the PMC passes ran on, so that a
    # later bench run only quotes it while it still dispatches the same one
    try:
        with open(os.path.join(out, "bench_write.json")) as fh:
            bw = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
        if "hbm_bytes_per_launch_clear" in res:
            res["traffic_record"] = {"kernel": bw["roofline"]["kernel"].split(" grid")[0], "frames": frames,
                                     "hbm_bytes_per_launch": res["hbm_bytes_per_launch_clear"],
                                     "write_size_kb": w, "fetch_size_kb_raw": f,
                                     "method": "separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes over bench.py; "
                                               "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)"}
    except Exception as e:  # noqa: BLE001
        res["traffic_record"] = f"unavailable: {e}"
    try:
        with open(os.path.join(out, "bench_under_rocprof.json")) as fh:
            line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        res["bench_under_rocprof"] = {"ms_per_step": b["ms_per_step"], "kernel_ms": b["roofline"].get("kernel_ms"),
                                      "kernel_ms_spaced_launches": b["roofline"].get("kernel_ms_spaced_launches"),
                                      "frac": b["roofline"]["frac"], "value": b["value"]}
    except Exception as e:  # noqa: BLE001
        res["bench_under_rocprof"] = f"unreadable: {e}"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
