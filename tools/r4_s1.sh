#!/bin/bash
# round 4, session 1: the fused emit || resolve launch (parity + bench, pipelined vs serial in one process) and the
# transposed-accumulator decode epilogue (FR_DECODE_STORE=1: parity + same-box A/B)
export TMPDIR=/tmp
O=gpurun_out/r4s1
mkdir -p $O
timeout 900 python -m pytest tests/test_pipelined_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_pipelined.log
FR_DECODE_STORE=1 timeout 600 python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_decode_tr.log
for i in 1 2; do
  timeout 300 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  FR_DECODE_STORE=1 timeout 300 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_tr_$i.json 2> $O/bench_tr_$i.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2> $O/bench_k20.err
BCMD="python3 bench.py --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- $BCMD > $O/prof_bench.log 2>&1
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/pytest_pipelined.log $O/pytest_decode_tr.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4s1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, 'serial', round(d.get('serial_plan_faces_per_s',0)), (d.get('parity') or {}).get('ok'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
cat $O/prof_bench/*/*kernel_stats.csv 2>/dev/null | head -12
