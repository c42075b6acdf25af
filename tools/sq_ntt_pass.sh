#!/bin/bash
# One rocprofv3 counter pass (SQ block) over the standalone 2^24-point NTT; usage: tools/sq_ntt_pass.sh <tag> [counters...]
tag=$1; shift
ctrs="${@:-SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS}"
out=gpurun_out/sqntt_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 tools/bench_msm.py --log-n 24 --ntt --reps 3 > $out/run.json 2> $out/run.err
python3 tools/pmc_summary.py $(find $out -name "*counter_collection.csv" | head -1) > gpurun_out/sqntt_$tag.txt
find $out -name "*.db" -delete
