#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5o
mkdir -p $O
timeout 600 python tools/r5_q30_check.py > $O/q30_check.log 2>&1; echo "check rc=$?"; tail -1 $O/q30_check.log
( cd /tmp; FR_DECODE_ARITH=q30l4 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --route serial --steps 20 --warmup 5 --repeats 3 --cpu-faces 0 --no-ops-surface --parity-faces 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r5o/prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
for rep in a b; do
python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 8 2>/dev/null | tail -1 > $O/bench_$rep.json
python - <<PY
import json
d=json.loads(open('$O/bench_$rep.json').read())
q=d['q30_inflight']
print('f32', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), '| q30l4', round(d['q30_inflight_faces_per_s']), round(q['ms_per_step']*1e3,2), 'serial', round(q['serial_plan_ms_per_step']*1e3,2), {k[:6]:round(v*1e3,1) for k,v in q['serial_leg_kernels_avg_ms'].items()}, q['parity']['ok'], d['north_star_40pct_of_8TBs'])
PY
done
