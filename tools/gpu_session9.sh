#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s9
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --cpu-faces 0 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
for f in ('bench','bench_k20'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('ops_surface_faces_per_s'), d['parity']['ok'])
    except Exception as e: print(f, 'ERR', e)
PY
