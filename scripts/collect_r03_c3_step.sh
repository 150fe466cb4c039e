#!/bin/bash
# rocprofv3 evidence for configs[3] as bench_configs.py runs it now (draw_targets_multiscale: box maps with the polyline sampler
# riding in the launch + point splat; the separate operators are timed in the same process): kernel trace + stats, HBM traffic
# (WRITE_SIZE / FETCH_SIZE in passes of their own) and the issue mix.  Counters only ever ride with --kernel-trace; the program
# after `--` is the interpreter itself.   bash scripts/collect_r03_c3_step.sh  ->  gpurun_out/r03/c3_step/
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03/c3_step"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/line.json" 2> "$OUT/trace.err"; echo "[r03] trace rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/line_write.json" 2> "$OUT/write.err"; echo "[r03] write rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/line_fetch.json" 2> "$OUT/fetch.err"; echo "[r03] fetch rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc/a" -o a -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/pmc_a.json" 2> "$OUT/pmc_a.err"; echo "[r03] pmc a rc=$?"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc/b" -o b -- python3 "$ROOT/scripts/bench_configs.py" 3 > "$OUT/pmc_b.json" 2> "$OUT/pmc_b.err"; echo "[r03] pmc b rc=$?"
python3 "$ROOT/scripts/summarise_pmc_extra.py" "$OUT/pmc" splat_multi_sampler_kernel splat_multi_kernel splat_points_multi_kernel polyline_kernel > "$OUT/pmc_issue_mix.json"
python3 - <<'PY'
import csv, collections, glob, os, json
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03/c3_step/"
res = {}
for sub, counter in (("write", "WRITE_SIZE"), ("fetch", "FETCH_SIZE")):
    f = glob.glob(out + sub + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ("splat_multi", "splat_points_multi_kernel", "polyline_kernel")):
            agg[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res.setdefault(k, {})[counter + "_KB"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, v in res.items():   # HBM bytes per launch: writes + 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md)
    v["hbm_bytes_per_launch"] = int(v.get("WRITE_SIZE_KB", 0) * 1024 + 2 * v.get("FETCH_SIZE_KB", 0) * 1024)
json.dump(res, open(out + "traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
f = glob.glob(out + "trace/**/*kernel_stats.csv", recursive=True)[0]
print(open(f).read()[:1500])
PY
