#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
ACCV_HIP_LIB=accv-lab_amd/accvlab/_amd_native/libaccv_hip_tune.so timeout -k 10 400 python scripts/lane_points_probe.py --rule 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/lane_splat_rule_after_batched_loads.log | cut -c1-200
