#!/bin/bash
# A/B of an environment switch on bench.py, interleaved on one box:  bash tools/ab_env.sh VAR val_a val_b [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
for r in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python3 bench.py --steps 15 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$v', round(d['value'] / 1e6, 2), 'M samples/s', round(d['ms_per_step'], 3), 'ms/step', {k[:22]: round(x, 2) for k, x in d['kernel_ms_per_step'].items()})"
  done
done
