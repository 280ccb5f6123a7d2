#!/bin/bash
# PMC view of the whole-net TRAINING kernels (storing forward, data-gradient chain, batched weight gradient) of one real net:
#   bash tools/profile_fused_step.sh <tag> [siren|gauss|relu|posenc]     (on the GPU box, from the repo root)
# three --pmc passes in runs of their own (matrix-pipe busy / waits; FETCH_SIZE; WRITE_SIZE: KiB per dispatch, FETCH_SIZE to be
# doubled on gfx950 -- MI355X guide); summary in gpurun_out/<tag>_pmc_summary.txt
set -e
TAG=${1:-r04_fused_step}
NET=${2:-siren}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/pmc1 -- python3 tools/sweep_one.py $NET > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2 -- python3 tools/sweep_one.py $NET > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3 -- python3 tools/sweep_one.py $NET > $OUT/pmc3.log 2>&1
python3 tools/pmc_summary.py $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 -name "*counter_collection.csv") > gpurun_out/${TAG}_pmc_summary.txt
grep -A12 "fused_bwd_kernel\|fused_fwd_kernel\|gemmx2_tn16_kernel\|final_fused_kernel" gpurun_out/${TAG}_pmc_summary.txt | head -90
