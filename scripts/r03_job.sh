#!/bin/bash
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03/c3_traffic"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/line_write.json" 2> "$OUT/write.err"; echo "[r03] write rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/line_fetch.json" 2> "$OUT/fetch.err"; echo "[r03] fetch rc=$?"
python3 - <<'PY'
import csv, collections, glob, os, json
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r03/c3_traffic/"
res={}
for sub,counter in (("write","WRITE_SIZE"),("fetch","FETCH_SIZE")):
    f=glob.glob(out+sub+"/**/*counter_collection.csv", recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==counter and any(k in r["Kernel_Name"] for k in ("splat_multi_kernel","splat_points_multi_kernel","polyline_kernel")):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        res.setdefault(k,{})[counter+"_KB"]=sum(v)/len(v); res[k]["launches"]=len(v)
print(json.dumps(res, indent=1))
json.dump(res, open(out+"summary.json","w"), indent=1)
PY
