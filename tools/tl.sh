#!/bin/bash
# Bench line and kernel timeline of one proof for the tree as built; usage: tools/tl.sh <tag> [noprofile] [extra bench.py arguments]
# (writes gpurun_out/b_<tag>.json, gpurun_out/tl_<tag>.txt)
tag=$1; noprof=$2; shift; shift
export TMPDIR=/tmp
timeout -k 5 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 "$@" > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/b_$tag.json')); print('$tag', d['value'], d['proof_verified'], d['stage_ms'], d['kernel_ms_per_proof'], d['roofline']['isolated']['avg_launch_ms'])"
if [ -z "$noprof" ] || [ "$noprof" = "-" ]; then
mkdir -p gpurun_out/tl_$tag && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 "$@" > /dev/null 2> gpurun_out/tl_$tag/err.txt
f=$(find gpurun_out/tl_$tag -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py $f 0.002 > gpurun_out/tl_$tag.txt; rm -rf gpurun_out/tl_$tag
head -1 gpurun_out/tl_$tag.txt
fi
