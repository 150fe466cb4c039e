#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 300 python scripts/targets_two_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/targets_two_stream_probe.log | cut -c1-300
