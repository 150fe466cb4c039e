#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
ACCV_HIP_LIB=$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so timeout -k 10 300 python scripts/lane_points_probe.py --sweep > "$OUT/lane_probe_sweep3.log" 2>&1; echo "[r03] sweep rc=$?"
cat "$OUT/lane_probe_sweep3.log"
