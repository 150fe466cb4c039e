#!/bin/bash
# Builds the working tree's libaccv_hip.so with extra -D switches as accv-lab_amd/accvlab/_amd_native/libaccv_hip_<name>.so
# (compile-time experiment variants for in-process A/B; never shipped): scripts/build_variant_lib.sh <name> [-DFOO ...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
OBJ="$ROOT/build/variant_$NAME"
rm -rf "$OBJ" && mkdir -p "$OBJ"
cd "$ROOT/accv-lab_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function -fno-gpu-rdc -pthread "$@" -c "$f" -o "$OBJ/${f%.hip}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "../accvlab/_amd_native/libaccv_hip_$NAME.so" "$OBJ"/*.o
echo "built libaccv_hip_$NAME.so with $*"
