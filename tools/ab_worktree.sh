#!/bin/bash
# Same-box A/B of the working tree against the COMMITTED library (how the second half of round 3 was measured; resolves
# +-0.2 us where cross-session comparisons are lost in the +-5 % spread between boxes):
#   git worktree add -f _old HEAD && (cd _old && python -c "import __graft_entry__ as g; g.build()")   # here, once per HEAD
#   gpurun -- 'bash tools/ab_worktree.sh 3'                                                              # on the GPU box
# (_old/ is git-ignored but travels with the gpurun snapshot; remove it with `git worktree remove --force _old`.)
# same-box A/B: committed library in _old vs the working tree, N rounds
N=${1:-3}
for i in $(seq $N); do for v in old new; do
  if [ $v = old ]; then (cd _old; python bench.py --cpu-faces 0 --no-ops-surface --no-rccl-selftest 2>/dev/null | tail -1 > ../gpurun_out/tmp_line.json); else python bench.py --cpu-faces 0 --no-ops-surface --no-rccl-selftest 2>/dev/null | tail -1 > gpurun_out/tmp_line.json; fi
  python -c "
import json; d=json.loads(open('gpurun_out/tmp_line.json').read()); print('$v', round(d['value']), round(d['ms_per_step']*1e3,2), 'serial', round(64e3/d['serial_plan_faces_per_s']*1e3,2) if d.get('serial_plan_faces_per_s') else None, 'q30', round(d['q30_inflight']['ms_per_step']*1e3,2) if isinstance(d.get('q30_inflight'),dict) else None, {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d['parity']['ok'])"
done; done
