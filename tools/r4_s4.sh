#!/bin/bash
# round 4, session 4: LDS vertex-window staging in the emit kernel (FR_EMIT_STAGE): parity suites + same-box A/B
export TMPDIR=/tmp
O=gpurun_out/r4s4
mkdir -p $O
timeout 1200 python -m pytest tests/test_render_gpu.py tests/test_fuzz_gpu.py tests/test_pipelined_gpu.py tests/test_pipeline_gpu.py tests/test_fused_layer_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    FR_EMIT_STAGE=$v timeout 300 python bench.py --route serial --cpu-faces 0 --no-ops-surface > $O/bench_stage${v}_$i.json 2> $O/bench_stage${v}_$i.err
  done
done
cat $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s4/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, (d.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
