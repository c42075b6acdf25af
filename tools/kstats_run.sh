#!/bin/bash
# Kernel statistics of the benchmark at one size for the tree as built; usage: tools/kstats_run.sh <tag> <log_gates>
tag=$1; lg=$2
export TMPDIR=/tmp
out=gpurun_out/ks_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -o ks -- python3 bench.py --log-gates $lg --steps 5 --warmup 2 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > $out/bench.json 2> $out/err.txt || exit 1
python3 tools/kstats.py $(find $out/d -name "*kernel_stats.csv" | head -1) > gpurun_out/ks_$tag.txt
rm -rf $out/d
head -16 gpurun_out/ks_$tag.txt
