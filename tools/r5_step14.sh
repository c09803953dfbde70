#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5n
mkdir -p $O
run() { # name env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 4 --q30-levels 0 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'min/max', round(d['ms_per_step_min']*1e3,1), round(d['ms_per_step_max']*1e3,1), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), 'in-region', {n:round(v.get('in_region_avg_ms',0)*1e3,1) for n,v in k.items() if n!='render_op'}, 'lat', round(1e3*d['per_batch_latency_ms'],1), (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
for rep in a b; do
run f32_free_$rep FR_DECODE_ARITH=f32 FR_INFLIGHT_ALTERNATE=0
run f32_alt_$rep FR_DECODE_ARITH=f32 FR_INFLIGHT_ALTERNATE=1
run q30_free_$rep FR_DECODE_ARITH=q30l4 FR_INFLIGHT_ALTERNATE=0
run q30_alt_$rep FR_DECODE_ARITH=q30l4 FR_INFLIGHT_ALTERNATE=1
done
