#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_15.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_15.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_15.log"; exit 1; }
timeout -k 10 300 python scripts/lane_raster_probe.py > "$OUT/lane_raster_probe.log" 2>&1; echo "[r03] raster probe rc=$?"; cat "$OUT/lane_raster_probe.log"
