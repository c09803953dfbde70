#!/usr/bin/env python3
"""emit_fixed_term_report.py TRACE_DIR OUT.json: raster_emit_kernel's duration by batch size from a rocprofv3 kernel trace of
tools/emit_fixed_term.py (grid = 210 workgroups per face), a least-squares line through the sizes of 16 faces and more, and the
staircase a workgroup-lifetime model predicts."""
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "raster_emit_kernel" not in r["Kernel_Name"]:
            continue
        wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256)
        grid = int(r.get("Grid_Size_X") or r.get("Grid_Size"))
        faces = grid // wg // 210
        rows.setdefault(faces, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
res = {}
for b in sorted(rows):
    v = sorted(rows[b][5:])        # the first launches of a size run on cold tables
    res[b] = {"launches": len(v), "median_us": round(v[len(v) // 2], 2), "min_us": round(v[0], 2), "max_us": round(v[-1], 2)}
xs = [b for b in res if b >= 16]
if len(xs) >= 2:
    n = len(xs)
    sx, sy = sum(xs), sum(res[b]["median_us"] for b in xs)
    sxx, sxy = sum(b * b for b in xs), sum(b * res[b]["median_us"] for b in xs)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    icpt = (sy - slope * sx) / n
else:
    slope = icpt = None
rec = {"what": "raster_emit_kernel under rocprofv3 --kernel-trace, one launch at a time on an idle chip, by batch size (210 workgroups of "
               "256 threads per face; 2,048 resident at a time)",
       "by_faces": res, "fit_16_faces_and_more": {"us_per_face": slope and round(slope, 4), "fixed_us": icpt and round(icpt, 2)},
       "workgroups_resident": 2048, "rounds_by_faces": {b: round(b * 210 / 2048, 2) for b in res}}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
