#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 600 python scripts/bench_secondary.py --configs F4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/secondary_f4_final.log | cut -c1-500
