#!/bin/bash
# round 4, session 2: transposed-accumulator decode epilogue (FR_DECODE_STORE=1) against the default, same box, interleaved
export TMPDIR=/tmp
O=gpurun_out/r4s2
mkdir -p $O
for i in 1 2 3; do
  for v in 0 1; do
    FR_DECODE_STORE=$v timeout 300 python bench.py --route serial --cpu-faces 0 --no-ops-surface > $O/bench_store${v}_$i.json 2> $O/bench_store${v}_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s2/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, (d.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
