#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_full.log" 2>&1; echo "rc=$?"; tail -2 "$OUT/gpu_tests_full.log" | cut -c1-200
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
