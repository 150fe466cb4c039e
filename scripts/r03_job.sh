#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
N=accv-lab_amd/accvlab/_amd_native
for args in "" "--nmin 128" "--rule B"; do
  tag=$(echo "$args" | tr -d ' -')
  timeout -k 10 300 python scripts/h1_variants.py --alt-lib $N/libaccv_hip_nopairs.so --rounds 5 $args > "$OUT/h1_ab_nopairs_occ8_$tag.log" 2>&1; echo "[r03] nopairs $args rc=$?"
  grep -E "lib" "$OUT/h1_ab_nopairs_occ8_$tag.log"
done
