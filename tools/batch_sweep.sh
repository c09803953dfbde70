#!/bin/bash
# Development record: the bench line at batch sizes either side of the configuration the metric is quoted on (64 faces).
# usage (GPU box): tools/batch_sweep.sh [outdir]
set -o pipefail
OUT=${1:-gpurun_out/r6sweep}
mkdir -p "$OUT"
for B in 16 32 48 64 128 256; do
  timeout 300 python3 bench.py --batch $B --steps 50 --warmup 10 --cpu-faces 0 --no-ops-surface --no-rccl-selftest \
      > "$OUT/b$B.json" 2> "$OUT/b$B.err" || echo "batch $B: rc $?" >> "$OUT/failures.txt"
done
python3 - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
rows = []
for B in (16, 32, 48, 64, 128, 256):
    p = os.path.join(out, "b%d.json" % B)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append({"batch": B, "error": repr(e)})
        continue
    q = d.get("q30_inflight") or {}
    rows.append({"batch": B, "faces_per_s": d["value"], "us_per_step": round(1e3 * d["ms_per_step"], 2),
                 "one_batch_at_a_time": d["config"].get("value_one_batch_at_a_time"),
                 "q30_faces_per_s": d.get("q30_inflight_faces_per_s"),
                 "kernels_us": {k: round(1e3 * v["avg_ms"], 2) for k, v in (d.get("kernels") or {}).items() if "avg_ms" in v},
                 "parity_ok": (d.get("parity") or {}).get("ok"), "faces_checked": (d.get("parity") or {}).get("faces_checked")})
json.dump(rows, open(os.path.join(out, "summary.json"), "w"), indent=1)
for r in rows:
    print(r)
PY
