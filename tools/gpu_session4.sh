#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s4
mkdir -p $O
for cfg in "0 0" "0 1" "0 2" "0 3" "1 1" "1 2" "0 0"; do
  set -- $cfg
  timeout 300 ./tools/decode_probe 64 53215 $1 1 0 $2 > $O/decode_quick_t$1_prio$2.json 2>> $O/decode_quick.err
  python - <<PY
import json
d=json.load(open('$O/decode_quick_t$1_prio$2.json'))
print('tiled=$1 prio=$2', {k:(v['nt_back_to_back'],v['nt_after_512MiB_flush'],v['cached_back_to_back']) for k,v in d['timing_us'].items()})
s=d['stamps'][0]; print('   span',s['kernel_span_us_realtime'],'clk',s['clock_GHz_median'],'store_ep',s['item_store_epilogue_cycles_per_item'],'util',s['matrix_pipe_utilisation_inside_window'],'win',s['simd_item_window_cycles'])
PY
done
