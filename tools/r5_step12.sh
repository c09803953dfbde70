#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
run() { # name env... -- flags
  local name=$1; shift
  env "$@" timeout 300 python bench.py --cpu-faces 0 --no-ops-surface --parity-faces 4 --q30-levels 0 $FLAGS > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(1e3*d['serial_plan']['ms_per_step'],2), (d.get('parity') or {}).get('ok'))
except Exception as e: print('$name','ERR',e)
PY
}
for rep in a b; do
FLAGS="" run ev10_$rep FR_BENCH_EV_EVERY=10
FLAGS="" run ev1000_$rep FR_BENCH_EV_EVERY=1000
FLAGS="--steps 20 --warmup 5" run k20_ev10_$rep FR_BENCH_EV_EVERY=10
FLAGS="--steps 20 --warmup 5" run k20_ev1000_$rep FR_BENCH_EV_EVERY=1000
done
