#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
for rep in 1 2; do
for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=2000"; do
env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('${v:-default}', 'frac', round(r['frac'],4), 'region', round(r['frac_region'],4), 'wall', round(r['frac_wall'],4), 'ms_per_step', round(d['ms_per_step'],5))"
done; done
