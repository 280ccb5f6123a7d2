mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest4.log)
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
cp gpurun_out/r03_pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err
timeout -k 10 300 python tools/bench_sweep.py > gpurun_out/r03_sweep.jsonl 2> gpurun_out/r03_sweep.err
bash tools/dp_overhead.sh > gpurun_out/r03_dp_overhead.txt 2>&1
WIRE_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 6 --steps 8 --warmup 2 > gpurun_out/r03_bench_gloo6.log 2>&1
tail -3 gpurun_out/r03_gputest4.log; cut -c1-400 gpurun_out/r03_bench_final.json; cat gpurun_out/r03_dp_overhead.txt; tail -3 gpurun_out/r03_bench_gloo6.log | cut -c1-600
