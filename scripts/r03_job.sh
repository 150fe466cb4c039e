#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_10.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_10.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_10.log"; exit 1; }
timeout -k 10 300 python scripts/tails_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/tails_probe2.log" 2>&1; echo "[r03] tails rc=$?"
grep mask_to "$OUT/tails_probe2.log"
timeout -k 10 300 python scripts/h2_bandwidth.py > "$OUT/h2_bandwidth2.log" 2>&1; echo "[r03] h2 rc=$?"
grep mask_to_indices "$OUT/h2_bandwidth2.log"
