export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_render_gpu.py tests/test_pipelined_gpu.py tests/test_pipeline_gpu.py tests/test_fused_layer_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
FR_EMIT_ORDER=1 timeout 1200 python -m pytest tests/test_render_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2 3; do for v in 0 -1 1; do FR_EMIT_ORDER=$v timeout 300 python bench.py --cpu-faces 0 --no-ops-surface 2>/dev/null | tail -1 > gpurun_out/tmp_$v.json; python -c "
import json; d=json.loads(open('gpurun_out/tmp_$v.json').read()); print($v, round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(x['avg_ms']*1e3,1) for k,x in d['kernels'].items()}, d['parity']['ok'])"; done; done
