#!/bin/bash
# Collect the rocprofv3 evidence for the headline kernel on a GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats  of `python3 bench.py`            -> per-kernel durations
#   2. --pmc WRITE_SIZE        of a short bench run (own pass)  -> HBM bytes written per launch
#   3. --pmc FETCH_SIZE        of a short bench run (own pass)  -> HBM bytes read per launch
# Counters are never combined with trace domains other than --kernel-trace (pool rule); the program after `--` is the
# interpreter itself.  Output: gpurun_out/prof_final/{trace,write,fetch}/ + summary.json (scripts/summarise_profiles.py).
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03/prof_final"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "[collect] kernel trace done" >&2
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "[collect] WRITE_SIZE pass done" >&2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
echo "[collect] FETCH_SIZE pass done" >&2
python3 "$ROOT/scripts/summarise_profiles.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
