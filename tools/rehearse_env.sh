#!/bin/bash
# One rank of <world> at <log_gates> under several environment settings, on ONE box: tools/rehearse_env.sh <world> <log_gates> <steps> "VAR=1" "" ...
w=$1; lg=$2; steps=$3; shift; shift; shift
for round in 1 2; do
  for e in "$@"; do
    env $e timeout -k 10 400 python3 tools/shard_rehearsal.py --world $w --log-gates $lg --steps $steps > gpurun_out/re_tmp.json 2> gpurun_out/re_tmp.err || { echo "[$e] failed"; tail -3 gpurun_out/re_tmp.err; exit 1; }
    echo "[$e] $round $(tail -1 gpurun_out/re_tmp.json)"
  done
done
