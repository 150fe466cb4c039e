#!/bin/bash
# rocprofv3 evidence for the SECONDARY configs: per-kernel durations of configs[3] / configs[2] and of the
# H2 byte movers, plus two PMC passes (issue mix, waits) over the configs[3] kernels.  Counters only ever ride with
# --kernel-trace (pool rule); the program after `--` is the interpreter itself.  Run through gpurun from the repo root:
#   bash scripts/collect_r03_configs.sh [tag]        -> gpurun_out/r03/<tag>/...
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${1:-configs}"
OUT="$ROOT/gpurun_out/r03/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c3_trace" -o t -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/c3_line.json" 2> "$OUT/c3_trace.err"
echo "[r03] configs[3] kernel trace done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c2_trace" -o t -- python3 "$ROOT/scripts/bench_configs.py" 2 > "$OUT/c2_line.json" 2> "$OUT/c2_trace.err"
echo "[r03] configs[2] kernel trace done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/h2_trace" -o t -- python3 "$ROOT/scripts/h2_bandwidth.py" > "$OUT/h2_bandwidth.log" 2> "$OUT/h2_trace.err"
echo "[r03] H2 kernel trace done" >&2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/c3_pmc/a" -o a -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/c3_pmc_a.json" 2> "$OUT/c3_pmc_a.err"
echo "[r03] configs[3] PMC pass A done" >&2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/c3_pmc/b" -o b -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/c3_pmc_b.json" 2> "$OUT/c3_pmc_b.err"
echo "[r03] configs[3] PMC pass B done" >&2
python3 "$ROOT/scripts/summarise_pmc_extra.py" "$OUT/c3_pmc" splat_multi_kernel splat_points_multi_kernel polyline_kernel group_boxes_kernel > "$OUT/c3_pmc_summary.json"
cat "$OUT/c3_pmc_summary.json"
