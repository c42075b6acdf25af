#!/bin/bash
# A/B of library builds on ONE box: tools/ab.sh <steps> <dirA> <dirB> ...   (directories under plonk_amd/ holding both libraries;
# "lib" is the tree's own build).  Runs the benchmark for each directory in turn, twice round, and prints ms per proof.
steps=$1; shift
for round in 1 2; do
  for d in "$@"; do
    PLK_LIB_DIR=$PWD/plonk_amd/$d timeout -k 5 200 python3 bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > gpurun_out/ab_${d}_$round.json 2> gpurun_out/ab_${d}_$round.err || { echo "$d failed"; tail -3 gpurun_out/ab_${d}_$round.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/ab_${d}_$round.json')); print('$d', $round, d['value'], d['proof_verified'], d['stage_ms'], 'iso', d['roofline']['isolated']['avg_launch_ms'])"
  done
done
