#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_full.log" 2>&1; rc=$?; tail -3 "$OUT/gpu_tests_full.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "[r03] smoke rc=$?"; tail -1 "$OUT/smoke.log"
timeout -k 10 300 python bench.py > "$OUT/bench_final.json" 2> "$OUT/bench_final.err"; echo "[r03] bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$OUT/bench_final_k20.json" 2> "$OUT/bench_final_k20.err"; echo "[r03] bench k20 rc=$?"
python - <<'PY'
import json
for f in ("bench_final","bench_final_k20"):
    d=json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
    c=d["secondary"]["configs"]
    print(f, round(d["value"]), round(d["roofline"]["frac"],3), round(d["roofline"].get("frac_wall"),3), {k:round(v.get("value"),1) for k,v in c.items()})
PY
