#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03/final"
mkdir -p "$OUT"
cd "$ROOT"
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_k20_$i.json" 2> "$OUT/bench_k20.err"; echo "[r03] bench k20 rc=$?"
python bench.py > "$OUT/bench_$i.json" 2> "$OUT/bench.err"; echo "[r03] bench rc=$?"
done
python - <<'PY'
import json, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03/final/"
for f in ("bench_k20_1.json","bench_1.json","bench_k20_2.json","bench_2.json"):
    d=json.loads([l for l in open(out+f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, round(d["value"]), "frac", round(r["frac"],4), round(r["frac_wall"],4), {k:round(v["value"],1) for k,v in d["secondary"]["configs"].items()}, round(d["secondary"]["configs_wall_s"],1))
PY
