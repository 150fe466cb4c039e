#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_copier_gpu.py tests/test_copier_cpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py -x -q -k "copier or h3 or pipeline or packed or combine" 2>&1 | tail -4 | cut -c1-300
timeout -k 10 300 python scripts/mtc_breakdown.py --tensors 10000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/mtc_breakdown_tree_recycling.log | cut -c1-200
timeout -k 10 300 python scripts/mtc_breakdown.py --tensors 10000 --background 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03/mtc_breakdown_tree_recycling.log | grep "total\|free\|get()" | cut -c1-200
timeout -k 10 300 python scripts/bench_configs.py 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d['ms_per_step'], json.dumps(d.get('secondary'))[:600])"
