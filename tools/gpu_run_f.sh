#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py tests/test_config1.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/pytest_gpu.log
python tools/decode_probe2.py > $O/decode_probe2.log 2>&1
tail -4 $O/pytest_gpu.log; cat $O/decode_probe2.log
