#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2o
mkdir -p $O
python -m pytest tests/test_render_gpu.py tests/test_fused_layer_gpu.py tests/test_pipeline_gpu.py tests/test_config1.py tests/test_callers.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_gpu.log
python tools/emit_probe.py > $O/emit_probe.log 2>&1
tail -3 $O/pytest_gpu.log; cat $O/emit_probe.log
