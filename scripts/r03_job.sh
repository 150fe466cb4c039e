#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_draw_heatmap_gpu.py tests/test_bench_contract_gpu.py tests/test_config_sizes_gpu.py -m gpu -x -q > "$OUT/rows_rule_tests.log" 2>&1; rc=$?; tail -6 "$OUT/rows_rule_tests.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --no-configs > "$OUT/bench_rows_rule.json" 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_rows_rule.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"])
p=d["secondary"]["strong_scaling_prediction_from_one_gpu"]["splits"]
for k,v in p.items(): print(k, v["frames_per_gpu"], round(v["ms_slowest_shard"]*1e3,2), round(v["predicted_speedup"],2))
PY
