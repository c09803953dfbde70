#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv of tools/graph_gap_probe.py: the LAST 40 steps are the graph replays, the 40 before them
the eager steps.  Per mode: median kernel durations, median gap decode->emit, emit->resolve (inside a step) and resolve->decode
(between steps), in us."""
import csv, glob, json, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fr::" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [("D" if "decode" in r["Kernel_Name"] else "E" if "emit" in r["Kernel_Name"] else "R", int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
steps = []
i = 0
while i + 2 < len(ev):
    if [e[0] for e in ev[i:i + 3]] == ["D", "E", "R"]:
        steps.append(ev[i:i + 3]); i += 3
    else:
        i += 1
def rep(ss):
    med = lambda xs: round(statistics.median(xs) / 1e3, 2)
    out = {"decode": med([s[0][2] - s[0][1] for s in ss]), "emit": med([s[1][2] - s[1][1] for s in ss]), "resolve": med([s[2][2] - s[2][1] for s in ss]),
           "gap_decode_to_emit": med([s[1][1] - s[0][2] for s in ss]), "gap_emit_to_resolve": med([s[2][1] - s[1][2] for s in ss]),
           "gap_resolve_to_next_decode": med([b[0][1] - a[2][2] for a, b in zip(ss, ss[1:])]),
           "step_period": med([b[0][1] - a[0][1] for a, b in zip(ss, ss[1:])])}
    return out
n = len(steps)
print(json.dumps({"steps_found": n, "eager (steps -80..-41)": rep(steps[-79:-41]), "graph replay (last 39)": rep(steps[-39:])}, indent=1))
