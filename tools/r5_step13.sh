#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5m
mkdir -p $O
for arith in f32 q30l4; do
( cd /tmp; FR_DECODE_ARITH=$arith rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$arith -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --repeats 3 --cpu-faces 0 --no-ops-surface --parity-faces 0 --no-serial-leg --q30-levels 0 > $GRAFT_REPO_ROOT/$O/trace_$arith.log 2>&1 )
python tools/inflight_trace.py $O/trace_$arith $O/timeline_$arith.json
tail -1 $O/trace_$arith.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arith', d['value'], d['ms_per_step'])"
find $O/trace_$arith -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
done
