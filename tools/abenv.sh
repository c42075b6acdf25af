#!/bin/bash
# A/B of environment settings on ONE box with the tree's own build: tools/abenv.sh <steps> "VAR=1" "" ...   (an empty string = no setting)
steps=$1; shift
for round in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 5 200 python3 bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > gpurun_out/abenv_${i}_$round.json 2> gpurun_out/abenv_${i}_$round.err || { echo "[$e] failed"; tail -3 gpurun_out/abenv_${i}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/abenv_${i}_$round.json')); print('[$e]', $round, d['value'], d['proof_verified'], d['stage_ms'], d['kernel_ms_per_proof'], 'iso', d['roofline']['isolated']['avg_launch_ms'])"
  done
done
