#!/bin/bash
# GPU call 2 of round 3: suite on the zero-first lane splat + widened fused loss, lane probe A/B against the previous build
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_2.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_2.log)"
[ $rc -ne 0 ] && { tail -40 "$OUT/gpu_tests_2.log"; exit 1; }
timeout -k 10 300 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_probe_ab.log" 2>&1; echo "[r03] probe rc=$?"
cat "$OUT/lane_probe_ab.log"
timeout -k 10 300 python scripts/bench_configs.py 3 > "$OUT/c3_line_2.json" 2>&1; echo "[r03] c3 rc=$?"
cat "$OUT/c3_line_2.json"
