#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"; cd "$ROOT"
ACCV_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/bench_2rank_rehearsal.json" 2> "$OUT/bench_2rank.err"; echo "[r03] 2-rank rc=$?"
ACCV_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
  bench.py --gpus 2 --steps 20 --warmup 5 --scaling strong > "$OUT/bench_2rank_rehearsal_strong.json" 2> "$OUT/bench_2rank_strong.err"; echo "[r03] 2-rank strong rc=$?"
python - <<'PY'
import json
for f in ("bench_2rank_rehearsal","bench_2rank_rehearsal_strong"):
    try:
        d=json.loads(open(f"gpurun_out/r03/{f}.json").read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["scaling"], round(d["value"]), d["ms_per_step"], list(d["secondary"].keys())[:8])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 "$OUT/bench_2rank.err" | cut -c1-200
