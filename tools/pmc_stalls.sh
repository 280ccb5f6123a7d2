#!/bin/bash
# Where do the waves of the GEMM kernels wait?  Three rocprofv3 --pmc passes over a short bench run (own runs, --pmc only):
# instruction-class activity and waits, FIFO-full stalls of the vector-memory / LDS paths, texture-addresser / cache activity.
#   bash tools/pmc_stalls.sh   (on the GPU box, from the repo root; summaries under gpurun_out/stalls_*)
set -e
OUT=gpurun_out/stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
# STALL_CMD="python3 tools/sweep_one.py siren" profiles another workload (round 4: the whole-net training kernels)
BENCH=${STALL_CMD:-"python3 bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline"}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/p3 -- $BENCH > $OUT/p3.log 2>&1 || echo "pass 3 failed (counter names)"
python3 tools/pmc_summary.py $(find $OUT/p1 $OUT/p2 $OUT/p3 -name "*counter_collection.csv") > gpurun_out/stalls_summary.txt
rm -rf $OUT
grep -A26 "${STALL_GREP:-gemmx2h_nt_kernel<2, 2\|gemmx2h_nt_kernel<1, 2, 4, 1, true\|gemmx2_tn16_kernel<4, 2, 1}" gpurun_out/stalls_summary.txt | head -120
