#!/bin/bash
# Round-6 session 6 (GPU box): the resolver's strip-height hint for plans in flight (bench A/B, three interleaved rounds), its
# tests, and the emit kernel's duration by batch size under rocprofv3.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6f}
mkdir -p $O
timeout 900 python -m pytest tests/test_inflight_gpu.py tests/test_pipeline_gpu.py tests/test_programs_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 > $O/pytest.log
echo "pytest rc=$?"; tail -3 $O/pytest.log
BF="--steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 4 --q30-parity-faces 2 --parity-faces 2"
for r in 1 2 3; do
  for rows in 0 -1 7 6; do
    timeout 400 python3 bench.py $BF --strip-rows $rows > $O/bench_rows${rows}_r$r.json 2> $O/bench_rows${rows}_r$r.err || echo "bench rows=$rows r=$r rc=$?"
  done
done
python3 - $O <<'PY'
import json, sys, os, glob
o = sys.argv[1]
for p in sorted(glob.glob(os.path.join(o, "bench_rows*.json"))):
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(p), "NO LINE"); continue
    r = d["roofline"]
    print(os.path.basename(p), d["config"]["resolver_strip_rows"], round(1e3 * d["ms_per_step"], 2), round(r["step"]["frac_of_8TBs"], 4),
          "serial", round(1e3 * r["step"]["one_batch_at_a_time"]["ms_per_step"], 2), "q30", round(1e3 * r["q30"]["ms_per_step"], 2), r["q30"]["parity_ok"], d["parity"]["ok"],
          {k: round(1e3 * v["in_region_avg_ms"], 1) for k, v in d["kernels"].items() if "in_region_avg_ms" in v})
PY
rocprofv3 --kernel-trace --output-format csv -d $O/emit_fixed -- python3 tools/emit_fixed_term.py > $O/emit_fixed.log 2>&1; echo "emit_fixed rc=$?"
python3 tools/emit_fixed_term_report.py $O/emit_fixed $O/emit_fixed_term.json > /dev/null; echo "report rc=$?"
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
python3 -c "
import json; d=json.load(open('$O/emit_fixed_term.json')); print({b:v['median_us'] for b,v in d['by_faces'].items()}); print(d['fit_16_faces_and_more'])"
