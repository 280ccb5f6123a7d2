set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt/on -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/kt/on.log 2>&1
cp $(find gpurun_out/kt/on -name "*kernel_stats.csv" | head -1) gpurun_out/kt_split_on.csv
export WIRE_SPLIT_OUT=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt/off -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/kt/off.log 2>&1
cp $(find gpurun_out/kt/off -name "*kernel_stats.csv" | head -1) gpurun_out/kt_split_off.csv
rm -rf gpurun_out/kt
