#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_4.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_4.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_4.log"; exit 1; }
ACCV_HIP_LIB=$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so timeout -k 10 300 python scripts/lane_points_probe.py --sweep > "$OUT/lane_probe_sweep.log" 2>&1; echo "[r03] sweep rc=$?"
cat "$OUT/lane_probe_sweep.log"
timeout -k 10 300 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_probe_ab_diet.log" 2>&1; echo "[r03] probe rc=$?"
head -c 2500 "$OUT/lane_probe_ab_diet.log"
timeout -k 10 300 python scripts/mtc_breakdown.py --tensors 528 --iters 200 > "$OUT/mtc_breakdown_528.log" 2>&1; echo "[r03] mtc rc=$?"
cat "$OUT/mtc_breakdown_528.log"
