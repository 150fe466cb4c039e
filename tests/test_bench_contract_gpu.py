"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline` and `cpu_baseline`.
Run in-process on a reduced step count (the product path and the oracle leg are the real ones)."""
import io
import json
import sys
from contextlib import redirect_stdout

import pytest

pytestmark = pytest.mark.gpu


def _run_bench(monkeypatch, *argv):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields(monkeypatch):
    d = _run_bench(monkeypatch, "--gpus", "1", "--steps", "20", "--warmup", "5", "--batch", "8")
    assert d["metric"].startswith("heatmap frames/sec") and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # (8 frames: the one-round launch takes 128 x 32 tiles on a 256-CU chip, tests/test_draw_heatmap_gpu.py)
    assert r["kernel"].startswith(("splat_kernel<PX=4,R=8,CLEAR=1,SM=0>", "splat_kernel<PX=4,R=16,CLEAR=1,SM=0>"))
    assert r["algorithmic_bytes"] == 8 * 1080 * 1920 * 4 + 12 * d["config"]["objects_per_gpu"] + 4 * 8
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02            # kernel time cannot exceed the wall time of a step
    assert r["kernel_ms"] <= r["timed_region_event_ms_per_step"] * 1.02 and "launch K" in r["kernel_ms_source"]
    assert r["traffic"] is None or r["traffic"] > 0              # quoted only for the matching instantiation at the bench batch
    assert 0 < r["kernel_ms_spaced_launches"] < 2 * r["kernel_ms"]
    assert r["trace_index"]["timed_region_launches"] == 20
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["single_thread"]["cores"] == 1 and cb["single_thread"]["value"] > 0
    assert "inplace" in d["secondary"] and "rule_B" in d["secondary"]
    # the PMC record is for 64-frame launches of the 128 x 16 instantiation
    assert r["traffic"] is None and ("frames per launch" in r["traffic_source"] or "this run dispatched" in r["traffic_source"])
    assert r["frac_wall"] <= r["frac_region"] * 1.02 <= r["frac"] * 1.05
    # both scaling modes are in the line: `value` is the weak one, the strong split is predicted from one GPU at N = 1
    sec = d["secondary"]
    assert abs(sec["weak_scaling"]["frames_per_s"] - d["value"]) <= 1e-9 * d["value"]
    splits = sec["strong_scaling_prediction_from_one_gpu"]["splits"]
    assert [splits[n]["frames_per_gpu"] for n in ("1", "2", "4", "8")] == [64, 32, 16, 8]
    assert splits["1"]["predicted_speedup"] == 1.0 and splits["8"]["predicted_speedup"] > 2.0
    # the other single-GPU BASELINE configs ride in the same line, each with its own roofline and CPU baseline
    cfg = sec["configs"]
    assert sorted(cfg) == ["configs[0]", "configs[2]", "configs[3]"]
    for name, line in cfg.items():
        assert name in line["config"]["workload"] and line["value"] > 0 and line["ms_per_step"] > 0
        assert line["roofline"]["bound"] in ("host", "pcie", "hbm") and line["cpu_baseline"]["value"] > 0
    assert cfg["configs[3]"]["secondary"]["lane_raster_only_ms"] > 0
    c3 = cfg["configs[3]"]["roofline"]
    assert c3["traffic"] is None or 0.9 * c3["algorithmic_bytes"] < c3["traffic"] < 1.2 * c3["algorithmic_bytes"], c3["traffic_source"]
    assert sec["configs_wall_s"] < 60


def test_bench_strong_mode_reports_the_c4_split(monkeypatch):
    d = _run_bench(monkeypatch, "--gpus", "1", "--steps", "20", "--warmup", "5", "--scaling", "strong", "--no-configs",
                   "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and "configs[4]" in d["config"]["workload"]
    st = d["secondary"]["strong_scaling"]
    assert st["total_frames"] == 64 and st["frames_per_rank"] == [64] and st["range_this_rank"] == [0, 64]
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"] and d["ms_per_step"] == st["ms_per_step"]
    assert st["kernel"].startswith("splat_kernel<PX=4,R=8,CLEAR=1,SM=0>")
    assert d["secondary"]["weak_scaling"]["frames_per_s"] > 0
