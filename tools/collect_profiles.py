#!/usr/bin/env python3
"""Copies the summaries of a GPU profile session (tools/gpu_run_final.sh -> gpurun_out/<session>/) into profiles/ under
round-named files and refreshes profiles/pmc_traffic.json (the per-launch HBM traffic bench.py reports).
Usage: python tools/collect_profiles.py gpurun_out/r2final round2"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def first(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None   # the most recent run (gpurun merges every session's files into the same tree)


def kernel_stats(name, out):
    f = first("%s/**/*kernel_stats.csv" % name)
    if not f:
        return None
    shutil.copy(f, os.path.join(P, out))
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    hot = [r for r in rows if "fr::" in r["Name"]]
    return {"total_gpu_ms": tot / 1e6, "hot_path_ms": sum(float(r["TotalDurationNs"]) for r in hot) / 1e6,
            "hot_path_share": sum(float(r["TotalDurationNs"]) for r in hot) / tot,
            "hot_path_kernels": {r["Name"].split("(")[0].replace("void ", ""): {"calls": int(r["Calls"]),
                                                                                "avg_us": float(r["AverageNs"]) / 1e3}
                                 for r in hot}}


for j in ("bench", "bench_again", "bench_k20", "bench_serial", "bench_rows10", "bench_graph", "bench_q30", "bench_q30l5", "bench_q30l4",
          "phase_test", "inflight_timeline", "pmc_summary_q30l4"):
    f = os.path.join(src, j + ".json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_%s.json" % (tag, j)))
kernel_stats("prof_bench", "%s_kernel_stats.csv" % tag)
kernel_stats("prof_bwd", "%s_decode_bwd_kernel_stats.csv" % tag)
kernel_stats("prof_inflight", "%s_inflight_kernel_stats.csv" % tag)
kernel_stats("prof_q30l4", "%s_q30l4_kernel_stats.csv" % tag)
for lg in ("bwd_probe.log", "bwd_ab.log", "legs.log", "kernel_timing.log"):
    f = os.path.join(src, lg)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_%s" % (tag, lg)))
# the bench line printed by the SAME process the kernel stats come from (profiled: lower clocks, per-launch overhead)
f = os.path.join(src, "prof_bench.log")
if os.path.exists(f):
    lines = [l for l in open(f, errors="replace") if l.startswith("{") and '"metric"' in l]
    if lines:
        open(os.path.join(P, "%s_bench_under_rocprofv3.json" % tag), "w").write(lines[-1])
callers = {}
for cfg, js in (("c3", "config3_fwd"), ("c4", "config4_train_shard"), ("c5", "config5_fine448_shard")):
    rec = {}
    f = os.path.join(src, js + ".json")
    if os.path.exists(f) and os.path.getsize(f):
        rec["line"] = json.load(open(f))
    ks = kernel_stats("prof_" + cfg, "%s_%s_kernel_stats.csv" % (tag, js))
    if ks:
        rec["rocprofv3_kernel_stats"] = ks
    callers[js] = rec
json.dump(callers, open(os.path.join(P, "%s_caller_configs.json" % tag), "w"), indent=1)
f = os.path.join(src, "pmc_summary.json")
if os.path.exists(f):
    pmc = json.load(open(f))
    pmc = {k: v for k, v in pmc.items() if v.get("SQ_WAVES", 1) or v.get("FETCH_SIZE", 0)}
    json.dump(pmc, open(os.path.join(P, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
    dec = next(v for k, v in pmc.items() if "decode_ring_kernel" in k)
    emit = next(v for k, v in pmc.items() if "raster_emit_kernel" in k)
    res = next(v for k, v in pmc.items() if "resolve_write_kernel" in k)
    # rocprofv3 kernel averages of the same command (kernel_stats.csv of the prof_bench leg)
    avg = {}
    f2 = first("prof_bench/**/*kernel_stats.csv")
    if f2:
        for r in csv.DictReader(open(f2)):
            for key, pat in (("decode", "decode_ring_kernel"), ("raster_emit", "raster_emit_kernel"), ("resolve_write", "resolve_write_kernel")):
                if pat in r["Name"]:
                    avg[key] = float(r["AverageNs"]) / 1e6
    cal = None
    f3 = os.path.join(src, "pmc_calibration.json")
    if os.path.exists(f3):
        cal = json.load(open(f3))
        json.dump(cal, open(os.path.join(P, "%s_pmc_calibration.json" % tag), "w"), indent=1)
    # ONE stated correction, calibrated on kernels of known traffic in each kernel's access shape (tools/pmc_calib.hip):
    # FETCH_SIZE reads 0.500 of the bytes of 16-byte-per-lane streaming reads (the decode's fragments: doubled) and the bytes
    # at face value for dword gathers and scattered 16-byte loads (emit, resolve: raw); WRITE_SIZE is exact for every store
    # shape tried.  KiB * 1024, per launch.
    def traffic(v, fetch_factor):
        return (fetch_factor * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    json.dump({
        "source": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- "
                  "python3 bench.py --route serial --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0 "
                  "--no-rccl-selftest --q30-levels 0, MI355X, %s kernels (profiles/%s_pmc.json); per launch.  bench.py copies these figures "
                  "into its line (`traffic`, `rocprofv3_avg_ms`): they are NOT measured by the run that prints the line" % (tag, tag),
        "source_file": "profiles/pmc_traffic.json <- profiles/%s_pmc.json" % tag,
        "kernel_stats_csv": "profiles/%s_kernel_stats.csv" % tag,
        "correction": "FETCH_SIZE x2 for the decode (16 B/lane streams are tallied at half), x1 for emit / resolve (gathers are "
                      "tallied in full); WRITE_SIZE exact -- calibrated in profiles/%s_pmc_calibration.json" % tag,
        "calibration": None if cal is None else {k: {"fetch_over_known": cal[k]["fetch_over_known"], "write_over_known": cal[k]["write_over_known"]}
                                                 for k in ("calib_read16", "calib_gather4", "calib_write16", "calib_write4_12")},
        "batch": 64,
        "kernels": {
            "decode": {"traffic_bytes_per_launch": traffic(dec, 2.0), "fetch_KiB": dec["FETCH_SIZE"], "write_KiB": dec["WRITE_SIZE"],
                       "rocprofv3_avg_ms": avg.get("decode")},
            "raster_emit": {"traffic_bytes_per_launch": traffic(emit, 1.0), "fetch_KiB": emit["FETCH_SIZE"], "write_KiB": emit["WRITE_SIZE"],
                            "rocprofv3_avg_ms": avg.get("raster_emit")},
            "resolve_write": {"traffic_bytes_per_launch": traffic(res, 1.0), "fetch_KiB": res["FETCH_SIZE"], "write_KiB": res["WRITE_SIZE"],
                              "rocprofv3_avg_ms": avg.get("resolve_write")},
        },
    }, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for extra in ("kernel_timing.log", "decode_breakdown.json", "emit_phase_account.json", "emit_ablate.json", "emit_fixed_term.json", "decode_stamps_by_xcd.json", "bwd_probe.log", "legs.log", "pytest_gpu.log"):
    f = os.path.join(src, extra)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_%s" % (tag, extra)))
for sfx in ("", "_q30", "_q30l5", "_q30l4"):
    f = os.path.join(ROOT, "gpurun_out", "parity_depth_vs_f64%s.json" % sfx)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, "%s_parity_depth_vs_f64%s.json" % (tag, sfx)))
print(sorted(os.listdir(P)))
