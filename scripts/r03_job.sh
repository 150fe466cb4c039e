#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 400 python scripts/launch_split_probe.py accv-lab_amd/accvlab/_amd_native/libaccv_hip_nopairs.so > "$OUT/small_launch_occupancy_probe.log" 2>&1; echo "[r03] rc=$?"
cat "$OUT/small_launch_occupancy_probe.log"
