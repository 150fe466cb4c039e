#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
T=/tmp/f3prof; mkdir -p $T/new $T/prev "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T/new -o t -- python3 "$ROOT/scripts/bench_secondary.py" --configs F3 > $T/new.log 2>&1; echo "new rc=$?"
export ACCV_HIP_LIB="$ROOT/accv-lab_amd/accvlab/_amd_native/libaccv_hip_prev.so"
rocprofv3 --kernel-trace --stats --output-format csv -d $T/prev -o t -- python3 "$ROOT/scripts/bench_secondary.py" --configs F3 > $T/prev.log 2>&1; echo "prev rc=$?"
python3 - <<'PY' | tee "$OUT/f3_kernel_durations_index_pair.log"
import csv, glob
for v in ("new", "prev"):
    f = glob.glob(f"/tmp/f3prof/{v}/**/*kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if "matched_reduce" in r["Name"]:
            print(v, r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
