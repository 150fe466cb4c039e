#!/bin/bash
# Extra PMC passes over the headline kernel (own runs, --kernel-trace only beside the counters: pool rule):
#   pass A: VALU / SALU / LDS instruction counts and wave cycles  -> issue mix and VALU share
#   pass B: LDS bank conflicts vs LDS active cycles, wait cycles  -> is LDS or waiting the limiter
# Output: gpurun_out/prof_pmc/{a,b}/ ; summarised by scripts/summarise_pmc_extra.py
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > "$OUT/bench_a.json" 2> "$OUT/a.err"
echo "[pmc] pass A done" >&2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > "$OUT/bench_b.json" 2> "$OUT/b.err"
echo "[pmc] pass B done" >&2
python3 "$ROOT/scripts/summarise_pmc_extra.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
