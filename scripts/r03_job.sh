#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests_full.log" 2>&1; rc=$?; tail -4 "$OUT/gpu_tests_full.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "[r03] smoke rc=$?"; tail -2 "$OUT/smoke.log"
timeout -k 10 300 python bench.py > "$OUT/bench_after_fused.json" 2> "$OUT/bench_after_fused.err"; echo "[r03] bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_after_fused.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"].get("frac_wall"))
c=d["secondary"]["configs"]
for k,v in c.items(): print(k, v.get("value"), v.get("unit"))
PY
cd /tmp && export TMPDIR=/tmp
P="$OUT/lane_fused_prof"; mkdir -p "$P"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o t -- python3 "$ROOT/scripts/lane_fused_profile_target.py" > "$P/trace.out" 2> "$P/trace.err"; echo "[r03] trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$P/pmc/a" -o a -- python3 "$ROOT/scripts/lane_fused_profile_target.py" > "$P/pmc_a.out" 2> "$P/pmc_a.err"; echo "[r03] pmc a rc=$?"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$P/pmc/b" -o b -- python3 "$ROOT/scripts/lane_fused_profile_target.py" > "$P/pmc_b.out" 2> "$P/pmc_b.err"; echo "[r03] pmc b rc=$?"
python3 "$ROOT/scripts/summarise_pmc_extra.py" "$P/pmc" lane_raster_multi_kernel splat_points_multi_kernel polyline_kernel > "$P/pmc_summary.json"; cat "$P/pmc_summary.json" | cut -c1-1500
find "$P/trace" -name "*kernel_stats.csv" | head -1 | xargs -r head -8 | cut -c1-200
