#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s5
mkdir -p $O
# binary N tiled perm prio
for cfg in "decode_probe 53215 0 0 0" "decode_probe_noslp 53215 0 0 0" "decode_probe 53216 0 0 2" "decode_probe_noslp 53216 0 0 2" "decode_probe_noslp 53215 2 1 2" "decode_probe 53215 0 0 0"; do
  set -- $cfg
  timeout 300 ./tools/$1 64 $2 $3 1 $4 $5 > $O/q_$1_$2_t$3_p$4_prio$5.json 2>> $O/decode_quick.err
  python - <<PY
import json
d=json.load(open('$O/q_$1_$2_t$3_p$4_prio$5.json'))
print('$1 N=$2 tiled=$3 perm=$4 prio=$5', {k[:12]:(v['nt_back_to_back'],v['nt_after_512MiB_flush'],v['cached_back_to_back']) for k,v in d['timing_us'].items()})
for s in d['stamps']: print('  ',s['name'][:30],'span',s['kernel_span_us_realtime'],'clk',s['clock_GHz_median'],'store_ep',s['item_store_epilogue_cycles_per_item']['median'],'util',s['matrix_pipe_utilisation_inside_window']['median'],'mfma_item',s['item_mfma_stream_cycles_per_item']['median'])
PY
done
