#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s8
mkdir -p $O
for late in 0 1 0 1; do
  timeout 300 ./tools/decode_probe 64 53215 1 1 1 0 $late > $O/q_late$late.json 2>> $O/err
  python - <<PY
import json
d=json.load(open('$O/q_late$late.json'))
print('late=$late', {k[:12]:(v['nt_back_to_back'],v['nt_after_512MiB_flush'],v['cached_back_to_back']) for k,v in d['timing_us'].items()})
for s in d['stamps']: print('  ',json.dumps({k:v for k,v in s.items() if k not in ('name','note')}))
PY
done
