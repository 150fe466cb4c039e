#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_3.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_3.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_3.log"; exit 1; }
timeout -k 10 300 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_probe_ab_strips.log" 2>&1; echo "[r03] probe rc=$?"
cat "$OUT/lane_probe_ab_strips.log"
timeout -k 10 300 python scripts/bench_configs.py 3 > "$OUT/c3_line_3.json" 2>&1; echo "[r03] c3 rc=$?"
cat "$OUT/c3_line_3.json"
timeout -k 10 300 python scripts/launch_split_probe.py > "$OUT/launch_split_probe.log" 2>&1; echo "[r03] split rc=$?"
cat "$OUT/launch_split_probe.log"
