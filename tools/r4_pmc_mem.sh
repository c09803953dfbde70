#!/bin/bash
# round 4: where the vector-memory pipeline's time goes in the three kernels of the serial step (TA / TCP / SQ-VMEM / TCC counters,
# one rocprofv3 --pmc pass per set) -> gpurun_out/r4pmc/pmc_mem_summary.json
export TMPDIR=/tmp
O=gpurun_out/r4pmc
mkdir -p $O
BCMD="python3 bench.py --route serial --steps 10 --warmup 3 --repeats 2 --cpu-faces 0 --no-ops-surface --parity-faces 0"
i=0
for set in \
  "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  "TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum" \
  "TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
  "GRBM_GUI_ACTIVE GRBM_TA_BUSY" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/p$i -- $BCMD > $O/p$i.log 2>&1 || echo "pass $i failed: $set" >> $O/fail.log
done
python tools/pmc_summary.py $O/pmc_mem_summary.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 $O/p8 $O/p9 $O/p10 > /dev/null 2> $O/summary.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/fail.log 2>/dev/null; tail -2 $O/summary.err
