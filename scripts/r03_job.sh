#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_sample.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_sample.json").read().strip().splitlines()[-1])
c=d["secondary"]["configs"]
print(round(d["value"]), round(d["roofline"]["frac"],3), round(d["roofline"]["frac_wall"],3), {k:round(v["value"],1) for k,v in c.items()})
PY
