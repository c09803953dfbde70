#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
run() { # name env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 8 --q30-levels 0 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if n!='render_op'}, (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
for rep in a b; do
run q30l4_plain_$rep FR_DECODE_ARITH=q30l4 FR_DECODE_STORE=0
run q30l4_wt_$rep FR_DECODE_ARITH=q30l4 FR_DECODE_STORE=2
run q30l4_nt_$rep FR_DECODE_ARITH=q30l4 FR_DECODE_STORE=3
done
run f32 FR_DECODE_ARITH=f32
