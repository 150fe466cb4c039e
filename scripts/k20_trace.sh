set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02/k20
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02/k20'
b=json.loads([l for l in open(out+'/bench.json').read().splitlines() if l.startswith('{')][-1])
idx=b['roofline']['trace_index']
rows=list(csv.DictReader(open(glob.glob(out+'/trace/**/*kernel_trace.csv',recursive=True)[0])))
rows=[r for r in rows if 'splat_kernel<4, 8, true' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
a=idx['timed_region_first_launch']
sel=rows[a-5:a+22]
prev_end=None
for i,r in enumerate(sel):
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3 if prev_end else 0
    print(a-5+i, 'dur_us', round((e-s)/1e3,1), 'gap_before_us', round(gap,1), '<-- timed' if a<=a-5+i<a+20 else '')
    prev_end=e
print('bench ms_per_step', b['ms_per_step'], 'kernel_ms', b['roofline']['kernel_ms'], 'frac', b['roofline']['frac'])
PY
