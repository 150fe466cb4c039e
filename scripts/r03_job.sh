#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_multiscale_gpu.py tests/test_targets_multiscale_gpu.py tests/test_fuzz_gpu.py tests/test_config_sizes_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout -k 10 300 python scripts/box_maps_floor_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400; done
timeout -k 10 200 python scripts/bench_configs.py 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']; print('step', round(d['ms_per_step']*1e3,2), d['value'], 'box', round(s['box_maps_only_ms']*1e3,2), 'lanes', round(s['lane_raster_only_ms']*1e3,2))"
