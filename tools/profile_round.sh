#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
# Kernel trace + stats of the bench command, then PMC passes (separate runs, --pmc only: MI355X guide / gpurun rules):
#   pass 1: matrix-pipe busy, active cycles, wave cycles / waits;  pass 2: FETCH_SIZE;  pass 3: WRITE_SIZE
# Summaries land in gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -e
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$STATS" gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- $BENCH > $OUT/pmc3.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv") > gpurun_out/${TAG}_pmc_summary.txt
# per-class traffic / matrix-pipe busy + the hash of the sources they belong to: what bench.py's roofline.traffic reads
python3 tools/pmc_traffic.py $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv") > gpurun_out/${TAG}_pmc_traffic.json
C1=$(find $OUT/pmc1 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_dispatch_table.py "$C1" gemm > gpurun_out/${TAG}_pmc_dispatch_table.txt
head -40 gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
grep -A12 "gemmx" gpurun_out/${TAG}_pmc_summary.txt | head -100
