#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
BCMD="python3 bench.py --route serial --steps 20 --warmup 5 --repeats 3 --cpu-faces 0 --no-ops-surface --parity-faces 0"
prof() { # name env
  local name=$1; shift
  ( export "$@"; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$name -- python3 $GRAFT_REPO_ROOT/bench.py --route serial --steps 20 --warmup 5 --repeats 3 --cpu-faces 0 --no-ops-surface --parity-faces 0 > $GRAFT_REPO_ROOT/$O/prof_$name.log 2>&1 )
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; head -8 $f | cut -d, -f1-4,6-8 | cut -c1-200
  find $O/prof_$name -name "*kernel_trace.csv" -delete; find $O/prof_$name -name "*.db" -delete
}
run() { # name env...
  local name=$1; shift
  env "$@" python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 4 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
prof f32 FR_DECODE_ARITH=f32
prof q30l4_s1 FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=1
prof q30l4_s3 FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=3
prof q30l5_s1 FR_DECODE_ARITH=q30l5 FR_Q30_SCHED=1
run f32 FR_DECODE_ARITH=f32
run q30l4_s1 FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=1
run q30l4_s3 FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=3
run f32_b FR_DECODE_ARITH=f32
run q30l4_s1_b FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=1
run q30l4_s3_b FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=3
