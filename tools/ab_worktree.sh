# same-box A/B: committed library in _old vs the working tree, N rounds
N=${1:-3}
for i in $(seq $N); do for v in old new; do
  if [ $v = old ]; then (cd _old; python bench.py --cpu-faces 0 --no-ops-surface 2>&1 | tail -1 > ../gpurun_out/tmp_line.json); else python bench.py --cpu-faces 0 --no-ops-surface 2>&1 | tail -1 > gpurun_out/tmp_line.json; fi
  python -c "
import json; d=json.loads(open('gpurun_out/tmp_line.json').read()); print('$v', round(d['value']), round(d['ms_per_step']*1e3,2), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d['parity']['ok'])"
done; done
