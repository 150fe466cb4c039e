#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
ACCV_FUZZ_SCALE=60 timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py tests/test_fuzz_cpu.py -m "gpu or not gpu" -q -p no:cacheprovider > "$OUT/fuzz_soak.log" 2>&1; rc=$?; echo "[r03] soak rc=$rc $(tail -1 $OUT/fuzz_soak.log)"
[ $rc -eq 0 ] || tail -80 "$OUT/fuzz_soak.log" | cut -c1-250
