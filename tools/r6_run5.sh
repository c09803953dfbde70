#!/bin/bash
# Round-6 session 5 (GPU box): the pruned tree (pipelined route, lean resolver and probe #ifdefs gone) -- whole GPU suite, smoke,
# the emit kernel's ablation ladder at 1 / 16 / 64 faces (its fixed term itemised: VERDICT r5 item 4), the bench line twice.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6e}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 > $O/pytest_gpu.log
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
HF="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17"
[ -f tools/libemit_probe.so ] || hipcc $HF -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared -o tools/libemit_probe.so tools/emit_probe.hip
timeout 600 python tools/emit_ablate.py 1 16 64 > $O/emit_ablate_by_batch.json 2> $O/emit_ablate.err; echo "ablate rc=$?"
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2>> $O/bench.err; echo "bench k20 rc=$?"
python3 - $O <<'PY'
import json, sys, os
o = sys.argv[1]
for f in ("bench", "bench_k20"):
    try:
        d = json.loads(open(os.path.join(o, f + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    r = d["roofline"]
    print(f, round(d["value"]), round(1e3 * d["ms_per_step"], 2), "step frac", round(r["step"]["frac_of_8TBs"], 4), "serial", r["step"]["one_batch_at_a_time"],
          "q30", r["q30"] and (round(1e3 * r["q30"]["ms_per_step"], 2), round(r["q30"]["frac_of_8TBs"], 4), r["q30"]["parity_ok"]), "clock", r["clock_GHz_held"],
          {k: round(1e3 * v["avg_ms"], 1) for k, v in r["kernels"].items()}, "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"], 1), d["parity"]["ok"])
print(open(os.path.join(o, "emit_ablate_by_batch.json")).read()[:3000])
PY
