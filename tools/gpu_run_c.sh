#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
python tools/mall_probe.py > $O/mall_probe.log 2>&1
python -m pytest tests/test_backward_gpu.py tests/test_config3_gpu.py tests/test_losses_gpu.py tests/test_pipeline_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_gpu.log
python tools/kernel_timing.py > $O/kernel_timing.log 2>&1
cat $O/mall_probe.log; tail -5 $O/pytest_gpu.log; cat $O/kernel_timing.log
