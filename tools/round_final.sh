#!/bin/bash
# End-of-round evidence on the GPU box (from the repo root):  bash tools/round_final.sh r03
#   GPU test-suite, smoke, the profile round (kernel stats + PMC passes -> pmc_traffic.json of THESE sources), the bench
#   line, the config sweep, the DP-plumbing overhead.  Results under gpurun_out/; copy what should be judged to profiles/.
TAG=${1:-r04}
mkdir -p gpurun_out
# a step that was killed at its limit ends the script: no further GPU step after a timeout
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out (rc=$rc): $*" | tee -a gpurun_out/${TAG}_round_final.log; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest_final.log 2>&1
step timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
step timeout -k 10 600 bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
step timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err
step timeout -k 10 300 python tools/bench_sweep.py > gpurun_out/${TAG}_sweep.jsonl 2> gpurun_out/${TAG}_sweep.err
step timeout -k 10 300 bash tools/dp_overhead.sh > gpurun_out/${TAG}_dp_overhead.txt 2>&1
tail -3 gpurun_out/${TAG}_gputest_final.log; tail -1 gpurun_out/${TAG}_smoke.log; cut -c1-300 gpurun_out/${TAG}_bench_final.json; cat gpurun_out/${TAG}_dp_overhead.txt
