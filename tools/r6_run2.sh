#!/bin/bash
# Round-6 session 2 (GPU box): the balanced tile walk (FR_DECODE_WALK) -- probe A/B with ring depth and priorities, stamps by XCD,
# bit-exactness (decode, Q30, pipeline, in-flight, fuzz), bench A/B on one box with the Q30 leg; basis policy "auto" at 32 / 48 faces.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6b}
mkdir -p $O
HF="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17"
[ -x tools/decode_probe ] || hipcc $HF -o tools/decode_probe tools/decode_probe.hip
timeout 900 ./tools/decode_probe 64 53215 1 2 1 0 1 > $O/decode_walk_ab.json 2> $O/decode_walk_ab.err
echo "probe rc=$?"
timeout 1200 python -m pytest tests/test_decode_gpu.py tests/test_decode_q30_gpu.py tests/test_pipeline_gpu.py tests/test_inflight_gpu.py tests/test_fuzz_gpu.py tests/test_fused_layer_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_decode.log
echo "pytest rc=$?"; tail -3 $O/pytest_decode.log
BF="--steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 4 --q30-parity-faces 2 --parity-faces 4"
for r in 1 2 3; do
  for w in 0 1; do
    FR_DECODE_WALK=$w timeout 400 python3 bench.py $BF > $O/bench_walk${w}_r$r.json 2> $O/bench_walk${w}_r$r.err || echo "bench walk=$w r=$r rc=$?"
  done
done
for B in 32 48; do
  timeout 300 python3 bench.py --batch $B --steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 0 --parity-faces 2 \
      > $O/policy_auto_b${B}.json 2> $O/policy_auto_b${B}.err || echo "policy B=$B rc=$?"
done
python3 - $O <<'PY'
import json, sys, os, glob
o = sys.argv[1]
def line(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return None
for p in sorted(glob.glob(os.path.join(o, "bench_walk*.json")) + glob.glob(os.path.join(o, "policy_*.json"))):
    d = line(p)
    if not d:
        print(os.path.basename(p), "NO LINE"); continue
    q = d.get("q30_inflight") or {}
    print(os.path.basename(p), round(d["value"]), round(1e3 * d["ms_per_step"], 2), round(d["config"].get("value_one_batch_at_a_time") or 0),
          {k: round(1e3 * v["avg_ms"], 1) for k, v in (d.get("kernels") or {}).items() if "avg_ms" in v}, (d.get("parity") or {}).get("ok"),
          "q30", q.get("ms_per_step") and round(1e3 * q["ms_per_step"], 2), q.get("serial_plan_ms_per_step") and round(1e3 * q["serial_plan_ms_per_step"], 2), (q.get("parity") or {}).get("ok"))
PY
