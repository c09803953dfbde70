#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
python tools/bwd_probe.py > $O/bwd_probe.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bwd -- python3 tools/bwd_probe.py > $O/prof_bwd.log 2>&1
python -m pytest tests/test_backward_gpu.py tests/test_config3_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_gpu.log
find $O -name "*kernel_trace.csv" -size +1M -delete
cat $O/bwd_probe.log; grep "bwd_\|render_backward" $O/prof_bwd/*/*kernel_stats.csv | tail -3 | cut -c1-160; tail -2 $O/pytest_gpu.log
