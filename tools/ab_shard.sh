#!/bin/bash
# A/B of library builds on one rank of a sharded proof, on ONE box: tools/ab_shard.sh <world> <log_gates> <steps> <dirA> <dirB> ...   (as tools/ab.sh)
w=$1; lg=$2; steps=$3; shift; shift; shift
for round in 1 2; do
  for d in "$@"; do
    PLK_LIB_DIR=$PWD/plonk_amd/$d timeout -k 10 400 python3 tools/shard_rehearsal.py --world $w --log-gates $lg --steps $steps > gpurun_out/abs${w}_${lg}_${d}_$round.json 2> gpurun_out/abs${w}_${lg}_${d}_$round.err || { echo "$d failed"; tail -3 gpurun_out/abs${w}_${lg}_${d}_$round.err; exit 1; }
    echo "$d $round $(tail -1 gpurun_out/abs${w}_${lg}_${d}_$round.json)"
  done
done
