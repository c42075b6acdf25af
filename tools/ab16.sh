#!/bin/bash
# A/B of library builds at another circuit size on ONE box: tools/ab16.sh <log_gates> <steps> <dirA> <dirB> ...   (as tools/ab.sh)
lg=$1; steps=$2; shift; shift
for round in 1 2; do
  for d in "$@"; do
    PLK_LIB_DIR=$PWD/plonk_amd/$d timeout -k 5 200 python3 bench.py --log-gates $lg --steps $steps --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > gpurun_out/ab${lg}_${d}_$round.json 2> gpurun_out/ab${lg}_${d}_$round.err || { echo "$d failed"; tail -3 gpurun_out/ab${lg}_${d}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/ab${lg}_${d}_$round.json')); print('$d', $lg, $round, d['value'], d['proof_verified'], d['stage_ms'], d['kernel_ms_per_proof'])"
  done
done
