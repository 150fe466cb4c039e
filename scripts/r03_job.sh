#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
timeout -k 10 600 python -m pytest tests/test_targets_multiscale_gpu.py tests/test_fuzz_gpu.py tests/test_pipeline_gpu.py tests/test_multiscale_gpu.py tests/test_lane_raster_gpu.py tests/test_lane_raster_fused_gpu.py tests/test_config_sizes_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python scripts/_host_cost.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
