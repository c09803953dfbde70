#!/bin/bash
# Round-6 session 3 (GPU box): the streaming decode on FEWER CUs than the chip has (FR_DECODE_CUS), two batches in flight: the
# decode's persistent workgroups take whole CUs, so today decode and the other batch's emit serialise chip-wide; with a grid
# below 256 the CUs left over run the other stream's emit / resolve workgroups for the whole decode.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6c}
mkdir -p $O
BF="--steps 100 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest --q30-levels 4 --q30-parity-faces 2 --parity-faces 2"
for r in 1 2; do
  for c in 0 240 224 208 192 176 160; do
    FR_DECODE_CUS=$c timeout 400 python3 bench.py $BF > $O/bench_cus${c}_r$r.json 2> $O/bench_cus${c}_r$r.err || echo "bench cus=$c r=$r rc=$?"
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
for p in sorted(glob.glob(os.path.join(o, "bench_cus*.json"))):
    d = line(p)
    if not d:
        print(os.path.basename(p), "NO LINE"); continue
    q = d.get("q30_inflight") or {}
    print(os.path.basename(p), round(d["value"]), round(1e3 * d["ms_per_step"], 2), round(d["config"].get("value_one_batch_at_a_time") or 0),
          {k: round(1e3 * v["avg_ms"], 1) for k, v in (d.get("kernels") or {}).items() if "avg_ms" in v}, (d.get("parity") or {}).get("ok"),
          "q30", q.get("ms_per_step") and round(1e3 * q["ms_per_step"], 2), q.get("serial_plan_ms_per_step") and round(1e3 * q["serial_plan_ms_per_step"], 2), (q.get("parity") or {}).get("ok"))
PY
