#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_13.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_13.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_13.log"; exit 1; }
timeout -k 10 900 bash scripts/collect_r03_configs.sh configs_final > "$OUT/collect_configs_final.log" 2>&1; echo "[r03] collect rc=$?"
tail -3 "$OUT/collect_configs_final.log"
python bench.py > "$OUT/bench_2.json" 2> "$OUT/bench_2.err"; echo "[r03] bench rc=$?"
