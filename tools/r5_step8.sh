#!/bin/bash
# how much of a change of the EMIT kernel's instruction count / duration reaches the in-flight step (VERDICT r4 item 2's premise)
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
run() { # name env...
  local name=$1; shift
  env "$@" python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 4 --q30-levels 0 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    k=d['kernels']
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if n!='render_op'}, 'in-region', {n:round(v.get('in_region_avg_ms',0)*1e3,1) for n,v in k.items() if n!='render_op'}, (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
for rep in a b; do for f in 3 1 0; do run filter${f}_$rep FR_EMIT_FILTER=$f; done; done
for rep in a b; do for o in 2 1 0; do run resolveopt${o}_$rep FR_RESOLVE_OPT=$o; done; done
