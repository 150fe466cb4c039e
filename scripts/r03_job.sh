#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_lane_raster_gpu.py tests/test_lane_raster_fused_gpu.py tests/test_targets_multiscale_gpu.py tests/test_config_sizes_gpu.py tests/test_draw_heatmap_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > "$OUT/walk_tests.log" 2>&1; rc=$?; tail -3 "$OUT/walk_tests.log" | cut -c1-300; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so > "$OUT/lane_points_probe_fixed_walk.log" 2>&1; echo "[r03] rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r03/lane_points_probe_fixed_walk.log'):
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['scales'], {k:(v['shipped']['us'],v['prev']['us'],v['prev'].get('same_as_shipped')) for k,v in d.items() if isinstance(v,dict) and 'shipped' in v})
PY
timeout -k 10 200 python scripts/bench_configs.py 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']; print('step', round(d['ms_per_step']*1e3,2), d['value'], 'box', round(s['box_maps_only_ms']*1e3,2), 'lanes', round(s['lane_raster_only_ms']*1e3,2))"
