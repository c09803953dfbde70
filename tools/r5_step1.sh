#!/bin/bash
# round 5, first session: i8 MFMA || VALU co-issue probe; the in-flight route with the Q30 decode as it stands
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
./tools/coexec3_probe > $O/coexec3.jsonl 2> $O/coexec3.err
python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_f32.json 2> $O/bench_f32.err
FR_DECODE_ARITH=q30 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_q30.json 2> $O/bench_q30.err
python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_f32_b.json 2>> $O/bench_f32.err
FR_DECODE_ARITH=q30 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_q30_b.json 2>> $O/bench_q30.err
cat $O/coexec3.jsonl
python - <<'PY'
import json
for f in ('bench_f32','bench_q30','bench_f32_b','bench_q30_b'):
    try:
        d=json.loads(open('gpurun_out/r5a/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(d.get('serial_plan_faces_per_s',0)), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, (d.get('parity') or {}).get('ok'))
    except Exception as e: print(f,'ERR',e)
PY
tail -3 $O/bench_q30.err
