#!/bin/bash
# round-2 GPU session A: parity suite, bench, configs 3-5 caller runs with rocprofv3 kernel stats
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/pytest_gpu.log
python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
python examples/coarse_loop.py --batch 32 --steps 5 > $O/config3_fwd.json 2> $O/config3_fwd.err
python examples/coarse_loop.py --batch 32 --steps 5 --train --val > $O/config4_train_shard.json 2> $O/config4_train_shard.err
python examples/coarse_loop.py --batch 16 --im-size 448 --steps 3 --fine > $O/config5_fine448_shard.json 2> $O/config5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 examples/coarse_loop.py --batch 32 --steps 3 > $O/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 examples/coarse_loop.py --batch 32 --steps 3 --train > $O/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 examples/coarse_loop.py --batch 16 --im-size 448 --steps 2 --fine > $O/prof_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 50 --warmup 5 --cpu-faces 0 > $O/prof_bench.log 2>&1
# keep only the stats summaries (the traces are large)
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
ls -R $O | head -50
