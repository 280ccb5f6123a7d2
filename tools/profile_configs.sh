#!/bin/bash
# rocprofv3 kernel stats of the other bench configurations (run on the GPU box from the repo root):
#   bash tools/profile_configs.sh r03     -> gpurun_out/<tag>_{siren,wire2d,relu,k181}_kernel_stats.csv
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in siren wire2d relu; do
  OUT=gpurun_out/prof_${TAG}_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/sweep_one.py $cfg > $OUT.log 2>&1
  cp "$(find $OUT -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_${cfg}_kernel_stats.csv
done
OUT=gpurun_out/prof_${TAG}_k181
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --hidden-features 256 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $OUT.log 2>&1
cp "$(find $OUT -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_k181_kernel_stats.csv
for cfg in siren wire2d relu k181; do echo "== $cfg"; head -6 gpurun_out/${TAG}_${cfg}_kernel_stats.csv | cut -c1-140; done
