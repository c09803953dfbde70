#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python tools/decode_probe2.py > $O/decode_probe2.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; python -c "
import json
d=json.load(open('$O/bench.json')); print(round(d['value']), d['ms_per_step'], d.get('value_min'), d.get('value_max'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}); print(d['cpu_baseline'])
"; head -3 $O/decode_probe2.log
