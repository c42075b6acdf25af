#!/bin/bash
# One rank of N at a given circuit size, N = 1, 2, 4, 8, on ONE GPU (tools/shard_rehearsal.py), then the kernel timeline of one proof
# of rank 0 of 8.  usage: tools/rehearse22.sh <log_gates> <tag>     (writes gpurun_out/rehearsal_<tag>.jsonl, gpurun_out/tl_<tag>_rank0_of_8.txt)
lg=$1; tag=$2
export TMPDIR=/tmp
: > gpurun_out/rehearsal_$tag.jsonl
for w in 1 2 4 8; do
  timeout -k 10 400 python3 tools/shard_rehearsal.py --world $w --log-gates $lg --steps 5 >> gpurun_out/rehearsal_$tag.jsonl 2> gpurun_out/rehearsal_$tag.err || { echo "world $w failed"; tail -3 gpurun_out/rehearsal_$tag.err; exit 1; }
  tail -1 gpurun_out/rehearsal_$tag.jsonl
done
mkdir -p gpurun_out/tlr_$tag && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlr_$tag -o tl -- python3 tools/shard_rehearsal.py --world 8 --log-gates $lg --steps 3 > /dev/null 2> gpurun_out/tlr_$tag/err.txt
f=$(find gpurun_out/tlr_$tag -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 0.002 0 > gpurun_out/tl_${tag}_rank0_of_8.txt; rm -rf gpurun_out/tlr_$tag
head -1 gpurun_out/tl_${tag}_rank0_of_8.txt
