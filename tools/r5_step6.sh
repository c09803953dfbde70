#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
run() { # name env...
  local name=$1; shift
  env "$@" python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 8 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if n!='render_op'}, 'in-region', {n:round(v.get('in_region_avg_ms',0)*1e3,1) for n,v in k.items() if n!='render_op'}, (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
run f32 FR_DECODE_ARITH=f32
for sc in 1 3 2; do run q30l4_s${sc} FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=$sc; done
run f32_b FR_DECODE_ARITH=f32
for sc in 1 3; do run q30l4_s${sc}_b FR_DECODE_ARITH=q30l4 FR_Q30_SCHED=$sc; done
