#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s12
mkdir -p $O
./tools/decode_probe 64 53215 1 1 1 1 | tee $O/ab_bal.json
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py tests/test_fuzz_gpu.py tests/test_decode_backward_gpu.py tests/test_losses_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --cpu-faces 0 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench.json')); print('bench', round(d['value']), d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('ops_surface_faces_per_s'), d['parity']['ok'])
PY
