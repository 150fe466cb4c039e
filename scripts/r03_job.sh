#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_lane_raster_fused_gpu.py tests/test_lane_raster_gpu.py tests/test_multiscale_gpu.py tests/test_config_sizes_gpu.py -m gpu -x -q > "$OUT/prologue_tests.log" 2>&1; rc=$?; tail -5 "$OUT/prologue_tests.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_points_probe_prologue.log" 2>&1; echo "[r03] rc=$?"
cat "$OUT/lane_points_probe_prologue.log" | cut -c1-400
timeout -k 10 300 python scripts/lane_fused_probe.py > "$OUT/lane_fused_probe4.log" 2>&1; echo "[r03] rc=$?"
cut -c1-330 "$OUT/lane_fused_probe4.log"
timeout -k 10 200 python scripts/bench_configs.py 3 > "$OUT/c3_prologue.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/c3_prologue.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d.get('breakdown', d.get('roofline')))[:600])"
