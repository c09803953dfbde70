#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3s6
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  timeout 300 ./tools/decode_probe 64 53215 $1 1 $2 > $O/q_pitch$1_prio$2.json 2>> $O/decode_quick.err
  python - <<PY
import json
d=json.load(open('$O/q_pitch$1_prio$2.json'))
print('pitched=$1 prio=$2', {k[:12]:(v['nt_back_to_back'],v['nt_after_512MiB_flush'],v['cached_back_to_back']) for k,v in d['timing_us'].items()})
for s in d['stamps']: print('  ',s['name'][:30],'span',s['kernel_span_us_realtime'],'clk',s['clock_GHz_median'],'prol',s['prologue_cycles']['median'],'prime',s['ring_prime_cycles']['median'],'store_ep',s['item_store_epilogue_cycles_per_item']['median'],'util',s['matrix_pipe_utilisation_inside_window']['median'],'win_us',s['simd_item_window_cycles']['median_us'],s['simd_item_window_cycles']['max_us'])
PY
done
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err
tail -3 $O/bench.err; python - <<PY
import json
for f in ('bench','bench_k20'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('ops_surface_faces_per_s'), d['parity']['ok'])
    except Exception as e: print(f, 'ERR', e)
PY
