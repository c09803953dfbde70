#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
python tools/emit_probe.py > $O/emit_probe.log 2>&1
python tools/kernel_timing.py > $O/kernel_timing.log 2>&1
python -m pytest tests/test_render_gpu.py tests/test_backward_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_gpu.log
cat $O/emit_probe.log $O/kernel_timing.log; tail -3 $O/pytest_gpu.log
