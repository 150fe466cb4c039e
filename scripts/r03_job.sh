#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_copier_gpu.py tests/test_copier_cpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py -x -q -k "copier or h3 or pipeline or packed or combine or recycling" 2>&1 | tail -3 | cut -c1-300
timeout -k 10 500 python scripts/bench_secondary.py --configs C2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/secondary_c2_final.log | cut -c1-400
