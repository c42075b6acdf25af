#!/bin/bash
# Sweep of environment settings at one circuit size on ONE box: tools/sweep16.sh <log_gates> <steps> "VAR=1 VAR2=2" "" ...
lg=$1; steps=$2; shift; shift
for round in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 5 200 python3 bench.py --log-gates $lg --steps $steps --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > gpurun_out/sw${lg}_${i}_$round.json 2> gpurun_out/sw${lg}_${i}_$round.err || { echo "[$e] failed"; tail -3 gpurun_out/sw${lg}_${i}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/sw${lg}_${i}_$round.json')); print('[$e]', $lg, $round, d['value'], d['proof_verified'], d['stage_ms'], d['kernel_ms_per_proof'])"
  done
done
