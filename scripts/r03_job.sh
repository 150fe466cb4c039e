#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
ACCV_HIP_LIB=accv-lab_amd/accvlab/_amd_native/libaccv_hip_boxr16.so timeout -k 10 300 python -m pytest tests/test_multiscale_gpu.py tests/test_targets_multiscale_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
for lib in "" accv-lab_amd/accvlab/_amd_native/libaccv_hip_boxr16.so; do
ACCV_HIP_LIB=$lib timeout -k 10 200 python scripts/bench_configs.py 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']; print('$lib'[-18:] or 'shipped', 'step', round(d['ms_per_step']*1e3,2), 'box', round(s['box_maps_only_ms']*1e3,2), 'lanes', round(s['lane_raster_only_ms']*1e3,2), 'separate', round(s['separate_operators_ms']*1e3,2))"
done; done
