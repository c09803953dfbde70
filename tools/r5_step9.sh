#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
set -o pipefail
python -m pytest tests/test_decode_backward_gpu.py tests/test_losses_gpu.py tests/test_config3_gpu.py tests/test_callers.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
python tools/decode_bwd_probe.py 2>&1 | grep "decode backward" | tee $O/bwd_probe.log
( cd /tmp; BWD_B=64 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bwd64 -- python3 $GRAFT_REPO_ROOT/tools/decode_bwd_probe.py > $GRAFT_REPO_ROOT/$O/prof_bwd64.log 2>&1 )
( cd /tmp; BWD_B=32 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bwd32 -- python3 $GRAFT_REPO_ROOT/tools/decode_bwd_probe.py > $GRAFT_REPO_ROOT/$O/prof_bwd32.log 2>&1 )
python - <<'PY'
import csv,glob
for d in ('prof_bwd64','prof_bwd32'):
    f=glob.glob('gpurun_out/r5i/%s/*/*kernel_stats.csv'%d)[0]
    print(d)
    for r in list(csv.DictReader(open(f)))[:6]:
        print('  ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
