#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_draw_heatmap_gpu.py tests/test_config_sizes_gpu.py tests/test_fuzz_gpu.py tests/test_multiscale_gpu.py -m gpu -x -q > "$OUT/ahead_tests.log" 2>&1; rc=$?; tail -3 "$OUT/ahead_tests.log"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/launch_split_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/small_launch_ahead_probe.log" 2>&1; echo "[r03] rc=$?"
cat "$OUT/small_launch_ahead_probe.log"
timeout -k 10 300 python bench.py > "$OUT/bench_ahead.json" 2> "$OUT/bench_ahead.err"; echo "[r03] bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_ahead.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("frac_wall"))
print(json.dumps(d["secondary"].get("strong_scaling_prediction_from_one_gpu"))[:600])
PY
