#!/bin/bash
# Round-6 session 7 (GPU box): the final binary under the widened differential fuzz (FR_FUZZ_CASES=6000), the batch sweep with the
# basis cache policy by batch, and the bench line once more with the driver's flags.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6g}
mkdir -p $O
FR_FUZZ_CASES=6000 timeout 2400 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $O/fuzz.log
echo "fuzz rc=$?"; tail -3 $O/fuzz.log
bash tools/batch_sweep.sh $O/sweep > $O/sweep.log 2>&1; echo "sweep rc=$?"; cat $O/sweep.log | tail -8
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench.err; echo "bench k20 rc=$?"
timeout 600 python3 bench.py --cpu-faces 0 --no-ops-surface > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?"
python3 - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for f in ("bench_k20", "bench"):
    d = json.loads(open(os.path.join(o, f + ".json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["value"]), round(1e3 * d["ms_per_step"], 2), round(r["step"]["frac_of_8TBs"], 4), "serial", round(1e3 * r["step"]["one_batch_at_a_time"]["ms_per_step"], 2),
          "q30", round(1e3 * r["q30"]["ms_per_step"], 2), round(r["q30"]["frac_of_8TBs"], 4), r["clock_GHz_held"], d["parity"]["ok"])
PY
