#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_full.log" 2>&1; rc=$?; tail -3 "$OUT/gpu_tests_full.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
ACCV_FUZZ_SCALE=10 timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q > "$OUT/fuzz_soak_final.log" 2>&1; echo "[r03] soak rc=$?"; tail -3 "$OUT/fuzz_soak_final.log" | cut -c1-300
