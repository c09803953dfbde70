#!/bin/bash
# round 4, session 3: XCD-contiguous decode walk (FR_DECODE_WALK=1) x segment-major emit map (FR_EMIT_MAP=1): parity + same-box A/B
export TMPDIR=/tmp
O=gpurun_out/r4s3
mkdir -p $O
FR_DECODE_WALK=1 FR_EMIT_MAP=1 timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py tests/test_render_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -4 > $O/pytest.log
for i in 1 2; do
  for v in "0 0" "1 0" "0 1" "1 1"; do
    set -- $v
    FR_DECODE_WALK=$1 FR_EMIT_MAP=$2 timeout 300 python bench.py --route serial --cpu-faces 0 --no-ops-surface > $O/bench_w$1_m$2_$i.json 2> $O/bench_w$1_m$2_$i.err
  done
done
cat $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s3/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, (d.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
