#!/usr/bin/env python3
"""Timeline of the in-flight route from a rocprofv3 --kernel-trace CSV: for the steady-state part of the run, per kernel kind,
start-to-start intervals, durations, how long each kernel ran ALONE vs beside a kernel of the other stream, and the device-idle
gaps.   python tools/inflight_trace.py <dir with *_kernel_trace.csv> [out.json]"""
import csv
import glob
import json
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    kind = "decode" if "decode_ring_kernel" in n or "decode_q_ring" in n else "stage" if "q_stage" in n else \
        "emit" if "raster_emit" in n else "resolve" if "resolve_write" in n else None
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# steady state: the last 60 % of the kernels
rows = rows[int(0.4 * len(rows)):]
t0, t1 = rows[0][0], rows[-1][1]
# sweep: time with 0 / 1 / 2+ kernels resident
ev = []
for s, e, k, q, st in rows:
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
active = {}
last = ev[0][0]
busy = {0: 0, 1: 0, 2: 0}
pair_time = {}
for t, dlt, k in ev:
    n = sum(active.values())
    busy[min(n, 2)] += t - last
    if n >= 1:
        key = "+".join(sorted(kk for kk, c in active.items() for _ in range(c)))
        pair_time[key] = pair_time.get(key, 0) + (t - last)
    last = t
    active[k] = active.get(k, 0) + dlt
per = {}
for s, e, k, q, st in rows:
    per.setdefault(k, []).append(e - s)
nb = len(per.get("resolve", []))
out = {"span_us": (t1 - t0) / 1e3, "batches": nb, "us_per_batch": (t1 - t0) / 1e3 / max(nb, 1),
       "device_time_us_per_batch": {"nothing_running": busy[0] / 1e3 / nb, "one_kernel": busy[1] / 1e3 / nb, "two_or_more": busy[2] / 1e3 / nb},
       "avg_duration_us": {k: sum(v) / len(v) / 1e3 for k, v in per.items()},
       "time_by_resident_set_us_per_batch": {k: v / 1e3 / nb for k, v in sorted(pair_time.items(), key=lambda kv: -kv[1])},
       "queues": sorted({q for *_, q, st in rows})}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
