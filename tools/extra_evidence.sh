#!/bin/bash
# extra evidence of the final tree: long soaks, kernel statistics of the 2^16 / 2^22 proofs and of the standalone 2^24-point MSM
set -e
export TMPDIR=/tmp
out=gpurun_out/extra_r02d
mkdir -p $out
timeout -k 5 200 python3 tools/soak.py --log-gates 20 --proofs 1000 > $out/soak20.txt 2>&1
timeout -k 5 200 python3 tools/soak.py --log-gates 16 --proofs 8000 > $out/soak16.txt 2>&1
timeout -k 5 200 python3 tools/soak.py --log-gates 10 --proofs 5000 > $out/soak10.txt 2>&1
for lg in 16 22; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench$lg -o b -- python3 bench.py --log-gates $lg --steps 3 --warmup 1 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > $out/bench$lg.json 2> $out/bench$lg.err
  python3 tools/kstats.py $(find $out/bench$lg -name "*kernel_stats.csv" | head -1) > $out/bench${lg}_kernel_stats.txt
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/msm24 -o b -- python3 tools/bench_msm.py --log-n 24 --reps 3 > $out/msm24.json 2> $out/msm24.err
python3 tools/kstats.py $(find $out/msm24 -name "*kernel_stats.csv" | head -1) > $out/msm24_kernel_stats.txt
find $out -name "*.db" -delete
find $out -name "*kernel_trace.csv" -delete
echo done
