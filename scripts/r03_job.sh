#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_targets_multiscale_gpu.py tests/test_multiscale_gpu.py tests/test_lane_raster_gpu.py tests/test_lane_raster_fused_gpu.py -m gpu -x -q > "$OUT/targets_tests.log" 2>&1; rc=$?; tail -25 "$OUT/targets_tests.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python scripts/bench_configs.py 3 > "$OUT/c3_targets.json" 2>/dev/null; python -c "
import json; d=json.loads(open('$OUT/c3_targets.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('secondary'))[:900])"
