#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_copier_gpu.py tests/test_copier_cpu.py tests/test_mtc_host_recycling_cpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py -x -q -k "copier or h3 or pipeline or packed or combine or recycling" 2>&1 | tail -4 | cut -c1-300
timeout -k 10 300 python scripts/bench_configs.py 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['value'], d['ms_per_step'], json.dumps(d.get('secondary'))[:600])"
