#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_18.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_18.log)"
[ $rc -eq 0 ] || { tail -60 "$OUT/gpu_tests_18.log"; exit 1; }
timeout -k 10 400 python scripts/bench_secondary.py --configs F4 > "$OUT/secondary_f4.log" 2>&1; echo "[r03] f4 rc=$?"; cat "$OUT/secondary_f4.log"
