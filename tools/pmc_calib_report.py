#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/pmc_calib's kernels (two rocprofv3 --pmc passes) against the bytes they are known to move.
usage: python tools/pmc_calib_report.py BYTES.json DIR_FETCH DIR_WRITE OUT.json"""
import csv, glob, json, os, sys
known = json.load(open(sys.argv[1]))
def counters(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            out.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}
c = {**counters(sys.argv[2]), **counters(sys.argv[3])}
rep = {"unit": "counter value (KiB) * 1024 / known bytes", "known_bytes": known}
for k, need in (("calib_read16", "calib_read16"), ("calib_gather4", "calib_gather4_unique"), ("calib_write16", "calib_write16"), ("calib_write4_12", "calib_write4_12")):
    f, w = c.get((k, "FETCH_SIZE")), c.get((k, "WRITE_SIZE"))
    rep[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
              "fetch_over_known": None if f is None or "write" in k else f * 1024 / known[need],
              "write_over_known": None if w is None or "write" not in k else w * 1024 / known[need]}
json.dump(rep, open(sys.argv[4], "w"), indent=1)
print(json.dumps(rep, indent=1))
