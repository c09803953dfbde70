#!/bin/bash
# round-3 profile session: parity suite, smoke, bench (default flags and the driver's K=20), rocprofv3 kernel stats and
# PMC passes of the same bench command, the decode / emit stamp probes, the decode backward under rocprofv3, caller configs
# 3-5 with kernel stats.  Everything lands under gpurun_out/r3final; tools/collect_profiles.py copies the summaries to
# profiles/round3_*.
export TMPDIR=/tmp
O=gpurun_out/r3final
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err
python bench.py --graph --cpu-faces 0 > $O/bench_graph.json 2>> $O/bench.err
FR_DECODE_ARITH=q30 python bench.py --cpu-faces 0 > $O/bench_q30.json 2>> $O/bench.err
BCMD="python3 bench.py --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- $BCMD > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- $BCMD > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- $BCMD > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- $BCMD > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $O/pmc_sq2 -- $BCMD > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -- $BCMD > $O/pmc_tcc.log 2>&1
python tools/pmc_summary.py $O/pmc_summary.json $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc > /dev/null 2>> $O/bench.err
python tools/kernel_timing.py > $O/kernel_timing.log 2>&1
./tools/decode_probe 64 53215 1 0 1 > $O/decode_breakdown.json 2> $O/decode_breakdown.err
./tools/decode_probe 64 53215 1 1 1 1 > $O/decode_ab.json 2>> $O/decode_breakdown.err
python tools/emit_probe.py > $O/emit_phase_account.json 2> $O/emit_probe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bwd -- python3 tools/decode_bwd_probe.py > $O/bwd_probe.log 2>&1
python examples/coarse_loop.py --config 3 --steps 5 > $O/config3_fwd.json 2> $O/config3_fwd.err
python examples/coarse_loop.py --config 4 --steps 5 > $O/config4_train_shard.json 2> $O/config4_train_shard.err
python examples/coarse_loop.py --config 5 --steps 3 > $O/config5_fine448_shard.json 2> $O/config5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 examples/coarse_loop.py --config 3 --steps 3 > $O/prof_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 examples/coarse_loop.py --config 4 --steps 3 > $O/prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -- python3 examples/coarse_loop.py --config 5 --steps 2 > $O/prof_c5.log 2>&1
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*.db" -delete
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; python -c "
import json
for f in ('bench','bench_k20','bench_graph','bench_q30'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], d.get('value_min'), d.get('value_max'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, d.get('graph_replay_faces_per_s'), d.get('ops_surface_faces_per_s'), (d.get('parity') or {}).get('ok'))
    except Exception as e: print(f, 'ERR', e)
"; cat $O/kernel_timing.log; cat $O/config3_fwd.json $O/config4_train_shard.json $O/config5_fine448_shard.json; grep "decode backward" $O/bwd_probe.log
