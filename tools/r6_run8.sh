#!/bin/bash
# Round-6 session 8 (GPU box): what a timed block's brackets cost at the driver's K = 20 -- the host waiting inside
# torch.cuda.synchronize() against polling hipEventQuery first -- and at K = 100; three interleaved rounds.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6h}
mkdir -p $O
BF="--cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 0 --parity-faces 2"
for r in 1 2 3; do
  for hw in sync poll; do
    for k in 20 100; do
      w=$((k / 4))
      timeout 400 python3 bench.py $BF --steps $k --warmup $w --host-wait $hw > $O/bench_${hw}_k${k}_r$r.json 2> $O/bench_${hw}_k${k}_r$r.err || echo "rc=$?"
    done
  done
done
python3 - $O <<'PY'
import json, sys, os, glob
o = sys.argv[1]
for p in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    b = sorted(d["blocks_ms_per_step"])
    print(os.path.basename(p), d["host_wait"], d["steps"], round(1e3 * d["ms_per_step"], 2), "serial", round(1e3 * d["roofline"]["step"]["one_batch_at_a_time"]["ms_per_step"], 2),
          "min", round(1e3 * b[0], 2), d["parity"]["ok"])
PY
