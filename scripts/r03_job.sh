#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
ACCV_HIP_LIB=$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so timeout -k 10 300 python scripts/lane_points_probe.py --rule > "$OUT/lane_probe_rule.log" 2>&1; echo "[r03] rc=$?"
cat "$OUT/lane_probe_rule.log"
timeout -k 10 300 python -m pytest tests/test_bool_indexing_bound.py tests/test_bench_contract_gpu.py -m gpu -q 2>&1 | tail -2
