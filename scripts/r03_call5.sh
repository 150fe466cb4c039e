#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 300 python -m pytest tests/test_lane_raster_gpu.py tests/test_multiscale_gpu.py tests/test_config_sizes_gpu.py -m gpu -x -q > "$OUT/gpu_tests_5.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_5.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_5.log"; exit 1; }
ACCV_HIP_LIB=$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so timeout -k 10 300 python scripts/lane_points_probe.py --sweep > "$OUT/lane_probe_sweep2.log" 2>&1; echo "[r03] sweep rc=$?"
cat "$OUT/lane_probe_sweep2.log"
