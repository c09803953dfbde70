#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
python tools/r5_q30_check.py > $O/q30_check.log 2>&1; echo "check rc=$?"
tail -5 $O/q30_check.log
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
run f32 FR_DECODE_ARITH=f32
for lv in 7 5 4; do for sc in 0 1 2; do run q30l${lv}_s${sc} FR_DECODE_ARITH=q30l${lv} FR_Q30_SCHED=$sc; done; done
run f32_b FR_DECODE_ARITH=f32
