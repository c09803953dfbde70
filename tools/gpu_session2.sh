#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s2
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/pytest_gpu.log
for cfg in "53215 0" "53216 0" "53215 1" "53215 0"; do
  set -- $cfg
  timeout 300 ./tools/decode_probe 64 $1 $2 1 > $O/decode_quick_N$1_t$2.json 2>> $O/decode_quick.err
  python - <<PY
import json
d=json.load(open('$O/decode_quick_N$1_t$2.json'))
print('N=$1 tiled=$2', {k:(v['nt_back_to_back'],v['nt_after_512MiB_flush'],v['cached_back_to_back']) for k,v in d['timing_us'].items()})
s=d['stamps'][0]; print('   span',s['kernel_span_us_realtime'],'clk',s['clock_GHz_median'],'store_ep',s['item_store_epilogue_cycles_per_item'],'util',s['matrix_pipe_utilisation_inside_window'],'win',s['simd_item_window_cycles'])
PY
done
tail -15 $O/pytest_gpu.log
