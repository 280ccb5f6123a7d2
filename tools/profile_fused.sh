#!/bin/bash
# rocprofv3 view of the whole-net forward kernel (wire_fused.hip) beside the layer-by-layer kernels it replaces:
#   bash tools/profile_fused.sh <tag> [net ...]      (on the GPU box, from the repo root; nets: tools/forward_only.py)
# kernel trace + stats, then PMC passes in runs of their own (matrix-pipe busy / waits / LDS conflicts; FETCH_SIZE; WRITE_SIZE:
# KiB per dispatch, FETCH_SIZE to be doubled on gfx950 -- MI355X guide); summaries land in gpurun_out/<tag>_*
set -e
TAG=${1:-r04_fused}
shift || true
NETS=${@:-siren}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export REPS=4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/forward_only.py $NETS > $OUT/trace.log 2>&1
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$STATS" gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/pmc1 -- python3 tools/forward_only.py $NETS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 tools/forward_only.py $NETS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 tools/forward_only.py $NETS > $OUT/pmc3.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv") > gpurun_out/${TAG}_pmc_summary.txt
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-220
grep -B1 -A10 "fused_fwd_kernel" gpurun_out/${TAG}_pmc_summary.txt | head -60
