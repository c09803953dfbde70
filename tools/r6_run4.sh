#!/bin/bash
# Round-6 session 4 (GPU box): (a) the streaming decode as TWO 8-wave workgroups per CU (FR_DECODE_WAVES=82): a decode workgroup then
# needs half a CU instead of a whole one, so with two batches in flight it can start beside the other batch's emit workgroups
# instead of behind them; (b) the new program-level tests + the bench tests (new roofline layout, N > 1 completeness).
export TMPDIR=/tmp
O=gpurun_out/${1:-r6d}
mkdir -p $O
BF="--steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 0 --parity-faces 2"
for r in 1 2 3; do
  for w in 16 82 8; do
    FR_DECODE_WAVES=$w timeout 400 python3 bench.py $BF > $O/bench_waves${w}_r$r.json 2> $O/bench_waves${w}_r$r.err || echo "bench waves=$w r=$r rc=$?"
  done
done
python3 - $O <<'PY'
import json, sys, os, glob
o = sys.argv[1]
def line(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return None
for p in sorted(glob.glob(os.path.join(o, "bench_waves*.json"))):
    d = line(p)
    if not d:
        print(os.path.basename(p), "NO LINE"); continue
    print(os.path.basename(p), round(d["value"]), round(1e3 * d["ms_per_step"], 2), round(d["config"].get("value_one_batch_at_a_time") or 0),
          {k: round(1e3 * v["avg_ms"], 1) for k, v in (d.get("kernels") or {}).items() if "avg_ms" in v},
          {k: round(1e3 * v["in_region_avg_ms"], 1) for k, v in (d.get("kernels") or {}).items() if "in_region_avg_ms" in v}, (d.get("parity") or {}).get("ok"),
          d["roofline"]["step"]["frac_of_8TBs"])
PY
timeout 2400 python -m pytest tests/test_programs_gpu.py tests/test_bench_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 > $O/pytest_programs.log
echo "pytest rc=$?"; tail -15 $O/pytest_programs.log
