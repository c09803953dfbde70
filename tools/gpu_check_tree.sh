#!/bin/bash
# (GPU box) The committed tree -- whole GPU suite, smoke, the bench line with the driver's flags and the default ones.
export TMPDIR=/tmp
O=gpurun_out/${1:-check}
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > $O/pytest_gpu.log
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench.err; echo "bench k20 rc=$?"
timeout 600 python3 bench.py > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?"
python3 - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for f in ("bench_k20", "bench"):
    d = json.loads(open(os.path.join(o, f + ".json")).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["value"]), round(1e3 * d["ms_per_step"], 2), round(r["step"]["frac_of_8TBs"], 4), "serial", round(1e3 * r["step"]["one_batch_at_a_time"]["ms_per_step"], 2),
          "q30", round(1e3 * r["q30"]["ms_per_step"], 2), round(r["q30"]["frac_of_8TBs"], 4), r["clock_GHz_held"], d["parity"]["ok"], "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"], 1),
          d["config"]["resolver_strip_rows"])
PY
