#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_gpu.log
python tools/overlap3_probe.py > $O/overlap3.log 2>&1
python examples/coarse_loop.py --batch 32 --steps 5 --train --val > $O/config4_train_shard.json 2> $O/config4_train_shard.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 examples/coarse_loop.py --batch 32 --steps 3 --train > $O/prof_c4.log 2>&1
python tools/kernel_timing.py > $O/kernel_timing.log 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
tail -5 $O/pytest_gpu.log; cat $O/overlap3.log | tail -5
