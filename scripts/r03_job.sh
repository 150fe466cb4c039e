#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
ACCV_FUZZ_SCALE=6 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q -k "lane_raster_paths or multiscale_and_lane" > gpurun_out/r03/fuzz_lane_paths.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r03/fuzz_lane_paths.log | cut -c1-400
