#!/bin/bash
mkdir -p gpurun_out/r2q
timeout 1500 python -m pytest tests/test_decode_q30_gpu.py tests/test_decode_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r2q/pytest.log 2>&1
tail -15 gpurun_out/r2q/pytest.log
