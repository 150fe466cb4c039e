#!/bin/bash
# evidence collection of round 3
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_final.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_final.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_final.log"; exit 1; }
ACCV_NO_FASTCALL=1 timeout -k 10 600 python -m pytest tests -m gpu -q > "$OUT/gpu_tests_ctypes_binding.log" 2>&1; echo "[r03] ctypes-binding suite rc=$? $(tail -1 $OUT/gpu_tests_ctypes_binding.log)"
ACCV_NO_HOST_FASTPATH=1 timeout -k 10 600 python -m pytest tests -m gpu -q > "$OUT/gpu_tests_python_formulations.log" 2>&1; echo "[r03] python-formulation suite rc=$? $(tail -1 $OUT/gpu_tests_python_formulations.log)"
timeout -k 10 300 python scripts/bench_secondary.py --configs F3 > "$OUT/secondary_f3.log" 2>&1; echo "[r03] f3 rc=$?"
cat "$OUT/secondary_f3.log"
timeout -k 10 600 bash scripts/collect_profiles.sh > "$OUT/collect_profiles.log" 2>&1; echo "[r03] collect_profiles rc=$?"
tail -5 "$OUT/collect_profiles.log"
