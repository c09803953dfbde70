#!/bin/bash
# Profile session of a round (run on the GPU box: gpurun -- 'bash tools/gpu_run_final.sh'): parity suite, smoke, bench (default
# flags, the driver's K = 20, hipGraph replay, Q30), rocprofv3 kernel stats and PMC passes of the same
# bench command, the PMC calibration kernels, the decode / emit probes, the decode backward under rocprofv3, caller configs
# 3-5 with kernel stats.  Everything lands under gpurun_out/$O; tools/collect_profiles.py copies the summaries to profiles/.
# Every leg's exit status is recorded in $O/legs.log; the script exits non-zero when any leg failed.
export TMPDIR=/tmp
O=gpurun_out/${1:-r6final}
mkdir -p $O
: > $O/legs.log
FAIL=0
leg() {   # leg NAME cmd...   (stdout -> $O/NAME.out unless the command redirects it itself)
  local name=$1; shift
  "$@"; local rc=$?
  echo "$name rc=$rc" >> $O/legs.log
  if [ $rc -ne 0 ]; then FAIL=1; echo "LEG FAILED: $name (rc=$rc)" >&2; fi
}
# ---- the probe binaries are git-ignored: build what is missing (the hipcc lines of the .hip headers) ------------------------
HF="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17"
[ -x tools/decode_probe ] || leg build_decode_probe hipcc $HF -o tools/decode_probe tools/decode_probe.hip
[ -f tools/libemit_probe.so ] || leg build_emit_probe hipcc $HF -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared -o tools/libemit_probe.so tools/emit_probe.hip
[ -x tools/pmc_calib ] || leg build_pmc_calib hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
# ---- parity, smoke, bench -------------------------------------------------------------------------------------------------
leg pytest_gpu bash -c "set -o pipefail; python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > $O/pytest_gpu.log"   # (pytest's own exit status: collection / fixture errors fail the leg too)
leg smoke bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
leg bench bash -c "python bench.py > $O/bench.json 2> $O/bench.err"
leg bench_k20 bash -c "python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err"
leg bench_serial bash -c "python bench.py --route serial --cpu-faces 0 --no-ops-surface > $O/bench_serial.json 2>> $O/bench.err"
leg bench_graph bash -c "python bench.py --graph --cpu-faces 0 --no-ops-surface > $O/bench_graph.json 2>> $O/bench.err"
leg bench_q30 bash -c "FR_DECODE_ARITH=q30 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_q30.json 2>> $O/bench.err"
leg bench_q30l5 bash -c "FR_DECODE_ARITH=q30l5 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_q30l5.json 2>> $O/bench.err"
leg bench_q30l4 bash -c "FR_DECODE_ARITH=q30l4 python bench.py --cpu-faces 0 --no-ops-surface > $O/bench_q30l4.json 2>> $O/bench.err"
leg bench_again bash -c "python bench.py --cpu-faces 0 > $O/bench_again.json 2>> $O/bench.err"
leg phase_test bash -c "python examples/coarse_loop.py --phase test --batch 32 --steps 6 --warmup 2 > $O/phase_test.json 2> $O/phase_test.err"
leg bench_rows10 bash -c "python bench.py --strip-rows 0 --cpu-faces 0 --no-ops-surface > $O/bench_rows10.json 2>> $O/bench.err"
# ---- rocprofv3: kernel stats + PMC passes of the same command (never --pmc together with a trace domain other than kernel-trace) --
# (BCMD = the serial route: one batch in flight, the state the line's per-kernel figures and roofline object are measured in;
#  prof_inflight = the default command, where kernels of two batches share the chip and a kernel's duration measures the sharing)
BCMD="python3 bench.py --route serial --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0 --no-rccl-selftest --q30-levels 0"
leg prof_bench bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- $BCMD > $O/prof_bench.log 2>&1"
leg prof_inflight bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_inflight -- python3 bench.py --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0 --no-serial-leg --no-rccl-selftest --q30-levels 0 > $O/prof_inflight.log 2>&1"
leg prof_q30l4 bash -c "FR_DECODE_ARITH=q30l4 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q30l4 -- $BCMD > $O/prof_q30l4.log 2>&1"
leg pmc_fetch_q30l4 bash -c "FR_DECODE_ARITH=q30l4 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch_q30l4 -- $BCMD > $O/pmc_fetch_q30l4.log 2>&1"
leg pmc_write_q30l4 bash -c "FR_DECODE_ARITH=q30l4 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write_q30l4 -- $BCMD > $O/pmc_write_q30l4.log 2>&1"
leg pmc_sq_q30l4 bash -c "FR_DECODE_ARITH=q30l4 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY -d $O/pmc_sq_q30l4 -- $BCMD > $O/pmc_sq_q30l4.log 2>&1"
leg pmc_summary_q30l4 bash -c "python tools/pmc_summary.py $O/pmc_summary_q30l4.json $O/pmc_fetch_q30l4 $O/pmc_write_q30l4 $O/pmc_sq_q30l4 > /dev/null 2>> $O/bench.err"
leg trace_inflight bash -c "rocprofv3 --kernel-trace --output-format csv -d $O/trace_inflight -- python3 bench.py --steps 100 --warmup 10 --repeats 3 --cpu-faces 0 --no-ops-surface --parity-faces 0 --no-serial-leg --q30-levels 0 --no-rccl-selftest > $O/trace_inflight.log 2>&1; python tools/inflight_trace.py $O/trace_inflight $O/inflight_timeline.json > /dev/null"
leg pmc_fetch bash -c "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- $BCMD > $O/pmc_fetch.log 2>&1"
leg pmc_write bash -c "rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- $BCMD > $O/pmc_write.log 2>&1"
leg pmc_sq1 bash -c "rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- $BCMD > $O/pmc_sq1.log 2>&1"
leg pmc_sq2 bash -c "rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $O/pmc_sq2 -- $BCMD > $O/pmc_sq2.log 2>&1"
leg pmc_tcc bash -c "rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -- $BCMD > $O/pmc_tcc.log 2>&1"
leg pmc_summary bash -c "python tools/pmc_summary.py $O/pmc_summary.json $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc > /dev/null 2>> $O/bench.err"
# ---- PMC calibration: FETCH_SIZE / WRITE_SIZE against kernels of known traffic in the three kernels' access shapes -----------
leg calib bash -c "./tools/pmc_calib > $O/calib_bytes.json && rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/calib_f -- ./tools/pmc_calib > /dev/null 2>&1 && rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/calib_w -- ./tools/pmc_calib > /dev/null 2>&1 && python tools/pmc_calib_report.py $O/calib_bytes.json $O/calib_f $O/calib_w $O/pmc_calibration.json > /dev/null"
# ---- probes ------------------------------------------------------------------------------------------------------------------
leg kernel_timing bash -c "python tools/kernel_timing.py > $O/kernel_timing.log 2>&1"
leg decode_breakdown bash -c "./tools/decode_probe 64 53215 1 0 1 > $O/decode_breakdown.json 2> $O/decode_breakdown.err"
leg emit_account bash -c "python tools/emit_probe.py > $O/emit_phase_account.json 2> $O/emit_probe.err"
leg emit_ablate bash -c "python tools/emit_ablate.py 16 64 > $O/emit_ablate.json 2> $O/emit_ablate.err"
leg emit_fixed bash -c "rocprofv3 --kernel-trace --output-format csv -d $O/emit_fixed -- python3 tools/emit_fixed_term.py > $O/emit_fixed.log 2>&1 && python3 tools/emit_fixed_term_report.py $O/emit_fixed $O/emit_fixed_term.json > /dev/null"
leg decode_xcd bash -c "./tools/decode_probe 64 53215 1 2 1 > $O/decode_stamps_by_xcd.json 2> $O/decode_stamps_by_xcd.err"
leg prof_bwd bash -c "BWD_B=64 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bwd -- python3 tools/decode_bwd_probe.py > $O/bwd_probe.log 2>&1"
leg bwd_probe bash -c "python tools/decode_bwd_probe.py >> $O/bwd_probe.log 2>&1"
leg bwd_ab bash -c "python tools/bwd_ab_probe.py > $O/bwd_ab.log 2>&1"
# ---- caller configs (BASELINE.json configs[2..4]) through bench.py --config N ----------------------------------------------------
leg config3 bash -c "python bench.py --config 3 --steps 5 > $O/config3_fwd.json 2> $O/config3_fwd.err"
leg config4 bash -c "python bench.py --config 4 --steps 5 > $O/config4_train_shard.json 2> $O/config4_train_shard.err"
leg config5 bash -c "python bench.py --config 5 --steps 3 > $O/config5_fine448_shard.json 2> $O/config5.err"
leg prof_c3 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 examples/coarse_loop.py --config 3 --steps 3 > $O/prof_c3.log 2>&1"
leg prof_c4 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 examples/coarse_loop.py --config 4 --steps 3 > $O/prof_c4.log 2>&1"
leg prof_c5 bash -c "rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 examples/coarse_loop.py --config 5 --steps 2 > $O/prof_c5.log 2>&1"
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/legs.log
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; python -c "
import json
for f in ('bench','bench_again','bench_k20','bench_rows10','bench_serial','bench_graph','bench_q30','bench_q30l5','bench_q30l4'):
    try:
        d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1]); print(f, round(d['value']), d['ms_per_step'], d.get('value_min'), d.get('value_max'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('graph_replay_faces_per_s'), d.get('ops_surface_faces_per_s'), d.get('serial_plan_faces_per_s'), (d.get('parity') or {}).get('ok'))
    except Exception as e: print(f, 'ERR', e)
"; cat $O/kernel_timing.log; cat $O/config3_fwd.json $O/config4_train_shard.json $O/config5_fine448_shard.json; grep "decode backward" $O/bwd_probe.log
exit $FAIL
