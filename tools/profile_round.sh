#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box: kernel statistics of the benchmark command, of the standalone
# 2^24 coset NTT and of the extended-coset quotient, and HBM traffic counters (separate passes, counters only).
# usage: tools/profile_round.sh <tag>      (run from the repository root; writes under gpurun_out/prof_<tag>/)
set -e
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
stats() {   # name, command...
    local name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o $name -- "$@" > $out/$name.json 2> $out/$name.err
    python3 tools/kstats.py $(find $out/$name -name "*kernel_stats.csv" | head -1) > $out/${name}_kernel_stats.txt
}
pmc() {     # name, counter, command...
    local name=$1 ctr=$2; shift 2
    timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d $out/${name}_$ctr -o p -- "$@" > /dev/null 2> $out/${name}_$ctr.err
    python3 tools/pmc_summary.py $(find $out/${name}_$ctr -name "*counter_collection.csv" | head -1) > $out/${name}_pmc_$ctr.txt
}
stats bench20 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-incl-h2d
stats ntt24 python3 tools/bench_msm.py --log-n 24 --reps 5 --ntt
stats quotient22 python3 tools/bench_msm.py --log-n 22 --reps 5 --quotient
for c in FETCH_SIZE WRITE_SIZE; do
    pmc bench20 $c python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0
    pmc ntt24 $c python3 tools/bench_msm.py --log-n 24 --reps 2 --ntt
    pmc quotient22 $c python3 tools/bench_msm.py --log-n 22 --reps 2 --quotient
done
# instruction-stream counters of the standalone 2^20-point MSM, and the figures bench.py quotes from these passes, tied to this build
tools/sq_pass.sh $tag
cp gpurun_out/sq_$tag.txt $out/msm20_pmc_sq.txt
python3 tools/profile_current.py $out/bench20_pmc_FETCH_SIZE.txt $out/bench20_pmc_WRITE_SIZE.txt $out/msm20_pmc_sq.txt $out/bench20.json plonk_amd/lib/libplonk_hip.so $tag 5 3 > $out/current.json
find $out -name "*.db" -delete
find $out -name "*kernel_trace.csv" -size +20M -delete
echo done
