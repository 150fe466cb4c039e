#!/bin/bash
# Builds libaccv_hip.so of an EARLIER commit (default HEAD) beside the working tree's, as
# accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so, for in-process A/B of code changes on one box
# (scripts/h1_variants.py --alt-lib, scripts/lane_points_probe.py --alt-lib).  Sources are exported into build/prev (ignored).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REV="${1:-HEAD}"
rm -rf "$ROOT/build/prev" && mkdir -p "$ROOT/build/prev"
git -C "$ROOT" archive "$REV" accv-lab_amd/csrc include | tar -x -C "$ROOT/build/prev"
make -C "$ROOT/build/prev/accv-lab_amd/csrc" -j8 OUT_DIR="$ROOT/build/prev/out" > "$ROOT/build/prev/build.log" 2>&1
cp "$ROOT/build/prev/out/libaccv_hip.so" "$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so"
echo "built $REV -> accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so"
