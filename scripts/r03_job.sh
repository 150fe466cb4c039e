#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_lane_raster_gpu.py tests/test_multiscale_gpu.py tests/test_config_sizes_gpu.py tests/test_fuzz_gpu.py tests/test_draw_heatmap_gpu.py -m gpu -x -q > "$OUT/gpu_tests_17.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_17.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_17.log"; exit 1; }
timeout -k 10 300 python scripts/lane_points_probe.py --brief --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_probe_inplace_prefetch.log" 2>&1; echo "[r03] probe rc=$?"
cat "$OUT/lane_probe_inplace_prefetch.log"
timeout -k 10 300 python scripts/small_splat_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/small_splat_probe2.log" 2>&1
cat "$OUT/small_splat_probe2.log"
