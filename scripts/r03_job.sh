#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
ACCV_FUZZ_SCALE=25 timeout -k 10 1000 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q -k "lane or multiscale or h1 or target or graph" > gpurun_out/r03/fuzz_soak_x25.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03/fuzz_soak_x25.log | cut -c1-300
