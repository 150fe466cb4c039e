#!/bin/bash
# First GPU call of round 3: suite, default bench line (with secondary.configs + strong-scaling prediction), 2-rank rehearsal
# of both scaling modes on one GPU (gloo), then the secondary-config profiles.
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1; echo "[r03] pytest rc=$? $(tail -1 $OUT/gpu_tests.log)"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "[r03] bench rc=$?"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_k20.json" 2> "$OUT/bench_k20.err"; echo "[r03] bench k20 rc=$?"
ACCV_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/bench_2rank_rehearsal.json" 2> "$OUT/bench_2rank.err"; echo "[r03] 2-rank rc=$?"
ACCV_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
  bench.py --gpus 2 --steps 20 --warmup 5 --scaling strong > "$OUT/bench_2rank_rehearsal_strong.json" 2> "$OUT/bench_2rank_strong.err"; echo "[r03] 2-rank strong rc=$?"
bash scripts/collect_r03_configs.sh configs > "$OUT/collect_configs.log" 2>&1; echo "[r03] collect rc=$?"
