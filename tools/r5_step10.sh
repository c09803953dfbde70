#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
set -o pipefail
python -m pytest tests/test_decode_backward_gpu.py tests/test_pipeline_gpu.py tests/test_inflight_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
python bench.py --graph --cpu-faces 0 --no-ops-surface --parity-faces 4 --q30-levels 0 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j/bench_graph.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'serial', round(d['serial_plan_faces_per_s']), 'graph', json.dumps(d.get('graph_replay'))[:700])
PY
