#!/bin/bash
# round-3 session 1: parity suite on the refactored library, decode cycle account, bench (new gate / legs)
export TMPDIR=/tmp
O=gpurun_out/r3s1
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -25 > $O/pytest_gpu.log
timeout 600 ./tools/decode_probe 64 > $O/decode_breakdown.json 2> $O/decode_breakdown.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err
tail -5 $O/pytest_gpu.log; head -c 3000 $O/decode_breakdown.json; tail -3 $O/bench.err; python - <<PY
import json
for f in ('bench','bench_k20'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('ops_surface_faces_per_s'), d.get('parity'))
    except Exception as e: print(f, 'ERR', e)
PY
