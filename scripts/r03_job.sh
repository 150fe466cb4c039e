#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p gpurun_out/r03
timeout -k 10 400 python scripts/lane_points_probe.py --alt-lib accv-lab_amd/accvlab/_amd_native/libaccv_hip_nolds.so > gpurun_out/r03/lane_points_probe_nolds.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03/lane_points_probe_nolds.log'):
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['scales'], {k:(v['shipped']['us'],v['nolds']['us']) for k,v in d.items() if isinstance(v,dict) and 'shipped' in v})
PY
