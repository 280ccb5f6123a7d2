#!/bin/bash
# Cost of the data-parallel plumbing on ONE GPU: bench.py with the RCCL all-reduce path kept live on a single rank
# (WIRE_DP_FORCE=1), collective issued from a side stream or from the compute stream, against the plain run.
#   bash tools/dp_overhead.sh        (from the repo root, on a GPU box)
pick() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'] / 1e6, 2), 'M samples/s', round(d['ms_per_step'], 3), 'ms/step')"; }
# default shape: per-layer slices on a side stream (wire_train_fwd_bwd_hooked + ncclAllReduce of parallel.RcclDirect)
for ov in layer none layer none; do
  WIRE_DP_OVERLAP=$ov WIRE_DP_FORCE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 1 --steps 15 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | pick "rccl-1rank direct, overlap=$ov"
done
# WIRE_DP_OVERLAP=none below: one all-reduce of the whole buffer after the backward
# (side_stream=0 with WIRE_DP_DIRECT=1: ncclAllReduce on the compute stream, parallel.RcclDirect;
#  WIRE_DP_DIRECT=0: torch.distributed.all_reduce of the process group)
export WIRE_DP_OVERLAP=none
for dd in 1 0 1 0; do
  WIRE_DP_DIRECT=$dd WIRE_DP_FORCE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 15 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | pick "rccl-1rank compute-stream direct=$dd"
done
for ss in 1 0 1 0; do
  WIRE_DP_SIDE_STREAM=$ss WIRE_DP_DIRECT=0 WIRE_DP_FORCE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 15 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | pick "rccl-1rank side_stream=$ss"
done
timeout -k 10 200 python3 bench.py --steps 15 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | pick "no collective"
