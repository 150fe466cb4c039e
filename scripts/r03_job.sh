#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
ACCV_NO_FASTCALL=1 timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests_ctypes_binding.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r03/gpu_tests_ctypes_binding.log | cut -c1-200
