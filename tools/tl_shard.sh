#!/bin/bash
# Kernel timeline of one proof of rank 0 of <world> at <log_gates>; usage: tools/tl_shard.sh <tag> <world> <log_gates>
tag=$1; w=$2; lg=$3
export TMPDIR=/tmp
mkdir -p gpurun_out/tlr_$tag && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlr_$tag -o tl -- python3 tools/shard_rehearsal.py --world $w --log-gates $lg --steps 3 > gpurun_out/tlr_$tag.json 2> gpurun_out/tlr_$tag/err.txt || exit 1
f=$(find gpurun_out/tlr_$tag -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 0.002 0 > gpurun_out/tl_${tag}.txt; rm -rf gpurun_out/tlr_$tag
head -1 gpurun_out/tl_${tag}.txt; tail -1 gpurun_out/tlr_$tag.json
