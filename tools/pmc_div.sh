#!/bin/bash
# SQ counters of the division kernels at 2^16 gates (one GPU); usage: tools/pmc_div.sh <tag>
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcdiv_$tag
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/pmcdiv_$tag -o p -- python3 bench.py --log-gates 16 --steps 2 --warmup 1 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > /dev/null 2> gpurun_out/pmcdiv_$tag.err
python3 tools/pmc_summary.py $(find gpurun_out/pmcdiv_$tag -name "*counter_collection.csv" | head -1) > gpurun_out/pmcdiv_$tag.txt
rm -rf gpurun_out/pmcdiv_$tag
grep "k_div\|k_msm_fold_tail\|marginals" gpurun_out/pmcdiv_$tag.txt
