#!/usr/bin/env bash
# AddressSanitizer / UBSan over the HOST code of this repo on the CPU test suite (GPU sanitizers are not available on the
# pool).  Two passes, because gcc's and clang's ASan run-times cannot share a process:
#   1. the g++-built torch extensions (_mtc_host, _bh_host, _lane_host, _dh_host, _fastcall) with -fsanitize=address,undefined
#   2. the host side of libaccv_hip.so (pack planner, pinned arena, host pack / polyline paths, argument checks) with
#      hipcc -Xarch_host -fsanitize=address, loaded through ACCV_HIP_LIB
# The in-tree binaries are put back afterwards.  Usage: scripts/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/accv-lab_amd"
TMP="$(mktemp -d /tmp/accv_sanitize.XXXXXX)"
EXTS=("$PKG"/accvlab/multi_tensor_copier/_mtc_host*.so "$PKG"/accvlab/batching_helpers/_bh_host*.so "$PKG"/accvlab/_amd_native/_fastcall*.so "$PKG"/accvlab/lane_helpers/polyline/_lane_host*.so "$PKG"/accvlab/draw_heatmap/_dh_host*.so)
mkdir -p "$TMP/keep" "$TMP/lib"
for f in "${EXTS[@]}"; do cp -p "$f" "$TMP/keep/"; done
restore() {
    cp -p "$TMP"/keep/_mtc_host*.so "$PKG/accvlab/multi_tensor_copier/"
    cp -p "$TMP"/keep/_bh_host*.so "$PKG/accvlab/batching_helpers/"
    cp -p "$TMP"/keep/_fastcall*.so "$PKG/accvlab/_amd_native/"
    cp -p "$TMP"/keep/_lane_host*.so "$PKG/accvlab/lane_helpers/polyline/"
    cp -p "$TMP"/keep/_dh_host*.so "$PKG/accvlab/draw_heatmap/"
    rm -rf "$TMP"
}
trap restore EXIT

echo "== pass 1: host extensions under gcc ASan + UBSan"
make -C "$PKG/csrc_host" clean >/dev/null
make -C "$PKG/csrc_host" CXX="g++ -fsanitize=address,undefined -fno-omit-frame-pointer -g" >/dev/null
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest "$ROOT/tests" -q -m "not gpu" -p no:cacheprovider "$@"
restore_only_exts() {
    cp -p "$TMP"/keep/_mtc_host*.so "$PKG/accvlab/multi_tensor_copier/"
    cp -p "$TMP"/keep/_bh_host*.so "$PKG/accvlab/batching_helpers/"
    cp -p "$TMP"/keep/_fastcall*.so "$PKG/accvlab/_amd_native/"
    cp -p "$TMP"/keep/_lane_host*.so "$PKG/accvlab/lane_helpers/polyline/"
    cp -p "$TMP"/keep/_dh_host*.so "$PKG/accvlab/draw_heatmap/"
}
restore_only_exts

echo "== pass 2: host side of libaccv_hip.so under clang ASan"
RT="$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)"
for f in accv_common draw_heatmap matched_loss polyline ragged_ops tensor_copier; do
    /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$PKG/csrc" -fno-gpu-rdc -pthread \
        -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c "$PKG/csrc/$f.hip" -o "$TMP/lib/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Xarch_host -fsanitize=address -shared-libsan \
    -o "$TMP/lib/libaccv_hip_asan.so" "$TMP"/lib/*.o
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 ACCV_HIP_LIB="$TMP/lib/libaccv_hip_asan.so" \
    python -m pytest "$ROOT/tests" -q -m "not gpu" -p no:cacheprovider "$@"
echo "== sanitizer passes clean"
