#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s14
mkdir -p $O
timeout 900 python -m pytest tests/test_decode_backward_gpu.py tests/test_fuzz_gpu.py tests/test_losses_gpu.py tests/test_config3_gpu.py tests/test_callers.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bwd -- python3 $GRAFT_REPO_ROOT/tools/decode_bwd_probe.py > $GRAFT_REPO_ROOT/$O/bwd_probe.log 2>&1; cd $GRAFT_REPO_ROOT
cat $O/bwd_probe.log | grep "decode backward"
find $O/prof_bwd -name "*kernel_stats.csv" | head -1 | xargs cat | head -12
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
