#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_lane_raster_fused_gpu.py tests/test_lane_raster_gpu.py -m gpu -x -q > "$OUT/fused_tests.log" 2>&1; rc=$?; tail -25 "$OUT/fused_tests.log" | cut -c1-400; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/lane_fused_probe.py > "$OUT/lane_fused_probe3.log" 2>&1; echo "[r03] rc=$?"
cat "$OUT/lane_fused_probe3.log"
