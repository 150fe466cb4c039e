#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python scripts/small_splat_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_smallpre.so > "$OUT/small_splat_preload_probe.log" 2>&1; echo "[r03] rc=$?"
cut -c1-600 "$OUT/small_splat_preload_probe.log"
