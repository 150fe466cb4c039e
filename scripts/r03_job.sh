#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 500 python scripts/bench_secondary.py --configs C2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/secondary_c2_final.log | cut -c1-400
