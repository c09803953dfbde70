#!/bin/bash
# round-2 session D: full parity suite, bench (default flags), rocprofv3 kernel stats + PMC passes of the same command
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --cpu-faces 0 > $O/bench_k20.json 2>> $O/bench.err
BCMD="python3 bench.py --steps 10 --warmup 3 --repeats 2 --cpu-faces 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- $BCMD > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -- $BCMD > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -- $BCMD > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc_sq1 -- $BCMD > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d $O/pmc_sq2 -- $BCMD > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -- $BCMD > $O/pmc_tcc.log 2>&1
python tools/pmc_summary.py $O/pmc_summary.json $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tcc > /dev/null 2>> $O/bench.err
python tools/kernel_timing.py > $O/kernel_timing.log 2>&1
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*.db" -delete
tail -4 $O/pytest_gpu.log; python -c "
import json
for f in ('bench','bench_k20'):
    d=json.load(open('$O/%s.json'%f)); print(f, round(d['value']), d['ms_per_step'], d.get('value_min'), d.get('value_max'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
"; cat $O/kernel_timing.log
