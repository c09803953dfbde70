#!/bin/bash
# Round-6 session 1 (GPU box): store-aware counted waits in the streaming decode (FR_DECODE_SAW) -- probe A/B + stamps by XCD,
# bit-exactness of the decode / pipeline tests, bench A/B on one box; then the basis cache policy by batch size (FR_DECODE_NT).
export TMPDIR=/tmp
O=gpurun_out/${1:-r6a}
mkdir -p $O
HF="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17"
[ -x tools/decode_probe ] || hipcc $HF -o tools/decode_probe tools/decode_probe.hip
timeout 600 ./tools/decode_probe 64 53215 1 2 1 0 1 > $O/decode_saw_ab.json 2> $O/decode_saw_ab.err
echo "probe rc=$?"
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_pipeline_gpu.py tests/test_inflight_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > $O/pytest_decode.log
echo "pytest rc=$?"; tail -3 $O/pytest_decode.log
BF="--steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 0 --parity-faces 4"
for r in 1 2 3; do
  for saw in 0 1; do
    FR_DECODE_SAW=$saw timeout 300 python3 bench.py $BF > $O/bench_saw${saw}_r$r.json 2> $O/bench_saw${saw}_r$r.err || echo "bench saw=$saw r=$r rc=$?"
  done
done
for B in 16 32 48 64; do
  for nt in 1 0; do
    FR_DECODE_NT=$nt timeout 300 python3 bench.py --batch $B --steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 0 --parity-faces 2 \
        > $O/policy_b${B}_nt$nt.json 2> $O/policy_b${B}_nt$nt.err || echo "policy B=$B nt=$nt rc=$?"
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
for p in sorted(glob.glob(os.path.join(o, "bench_saw*.json")) + glob.glob(os.path.join(o, "policy_*.json"))):
    d = line(p)
    if not d:
        print(os.path.basename(p), "NO LINE"); continue
    print(os.path.basename(p), round(d["value"]), round(1e3 * d["ms_per_step"], 2), d["config"].get("value_one_batch_at_a_time"),
          {k: round(1e3 * v["avg_ms"], 1) for k, v in (d.get("kernels") or {}).items() if "avg_ms" in v}, (d.get("parity") or {}).get("ok"))
PY
