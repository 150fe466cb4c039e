#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_11.log" 2>&1; rc=$?; echo "[r03] pytest rc=$rc $(tail -1 $OUT/gpu_tests_11.log)"
[ $rc -ne 0 ] && { tail -60 "$OUT/gpu_tests_11.log"; exit 1; }
for n in 10000 528; do for bg in "" "--background"; do
  timeout -k 10 300 python scripts/mtc_breakdown.py --tensors $n --iters 50 $bg > "$OUT/mtc_breakdown_recycle_${n}${bg}.log" 2>&1; echo "[r03] mtc $n $bg rc=$?"
  grep -E "tensors|total ms|free the|make_packed|rebuild \(|get\(\)" "$OUT/mtc_breakdown_recycle_${n}${bg}.log"
done; done
timeout -k 10 300 python scripts/bench_configs.py 2 > "$OUT/c2_line_recycle.json" 2>&1; echo "[r03] c2 rc=$?"
cat "$OUT/c2_line_recycle.json" | cut -c1-1500
