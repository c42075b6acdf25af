#!/bin/bash
# Kernel timelines of one proof at 2^16 gates (one GPU) and of rank 0 of 8 at 2^20 gates; usage: tools/tl_both.sh <tag>
tag=$1
bash tools/tl.sh ${tag}_16 - --log-gates 16 || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/tlr_$tag && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlr_$tag -o tl -- python3 tools/shard_rehearsal.py --world 8 --log-gates 20 --steps 3 > gpurun_out/tlr_$tag.json 2> gpurun_out/tlr_$tag/err.txt || exit 1
f=$(find gpurun_out/tlr_$tag -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 0.002 0 > gpurun_out/tl_${tag}_rank0_of_8.txt; rm -rf gpurun_out/tlr_$tag
head -1 gpurun_out/tl_${tag}_rank0_of_8.txt; tail -1 gpurun_out/tlr_$tag.json
