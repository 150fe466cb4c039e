#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_lane_helpers.py tests/test_lane_raster_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > "$OUT/gpu_tests_19.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_19.log)"
[ $rc -eq 0 ] || { tail -60 "$OUT/gpu_tests_19.log"; exit 1; }
timeout -k 10 300 python scripts/tails_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/tails_probe4.log" 2>&1; echo "[r03] rc=$?"
grep -E "polyline" "$OUT/tails_probe4.log"
