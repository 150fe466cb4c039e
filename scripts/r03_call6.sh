#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
N=accv-lab_amd/accvlab/_amd_native
timeout -k 10 300 python -m pytest tests/test_lane_raster_gpu.py tests/test_multiscale_gpu.py tests/test_config_sizes_gpu.py -m gpu -q --deselect tests/test_lane_raster_gpu.py::test_lane_splat_one_and_four_waves_per_tile_agree > "$OUT/gpu_tests_6.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_6.log)"
tail -3 "$OUT/gpu_tests_6.log"
timeout -k 10 300 python scripts/lane_points_probe.py --brief --alt-lib $N/libaccv_hip_prev.so $N/libaccv_hip_rowwalk.so $N/libaccv_hip_divcull.so > "$OUT/lane_probe_variants.log" 2>&1; echo "[r03] probe rc=$?"
cat "$OUT/lane_probe_variants.log"
