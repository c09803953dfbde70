#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
set -o pipefail
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_gpu.log; echo "pytest rc=$?"
tail -6 $O/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --cpu-faces 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5g/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value']), d['ms_per_step'], 'latency', d.get('per_batch_latency_ms'), 'clock', d.get('clock_GHz_held'))
print('one at a time', d['config'].get('value_one_batch_at_a_time'), 'q30', d.get('q30_inflight_faces_per_s'), json.dumps(d.get('q30_inflight'))[:900])
print('vector_pipe', json.dumps(d.get('vector_pipe'))[:400])
PY
python examples/coarse_loop.py --phase test --batch 32 --steps 6 --warmup 2 > $O/phase_test.json 2> $O/phase_test.err; echo "phase test rc=$?"; cat $O/phase_test.json | cut -c1-600; tail -2 $O/phase_test.err
