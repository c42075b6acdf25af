#!/bin/bash
# Bench lines, rehearsals and timelines that the round's notes quote; usage: tools/round_evidence.sh <tag>   (writes under gpurun_out/ev_<tag>/)
tag=$1
out=gpurun_out/ev_$tag
mkdir -p $out
export TMPDIR=/tmp
for lg in 16 22; do
  timeout -k 10 500 python3 bench.py --log-gates $lg --steps 10 --warmup 3 --no-cpu-baseline --ntt-log-n 0 > $out/bench$lg.json 2> $out/bench$lg.err || { echo "bench $lg failed"; tail -3 $out/bench$lg.err; exit 1; }
  python3 -c "import json; d=json.load(open('$out/bench$lg.json')); print($lg, d['value'], d['stage_ms'], d['kernel_ms_per_proof'])"
done
for lg in 20 22; do
  : > $out/rehearsal_2p$lg.jsonl
  for w in 1 2 4 8; do
    timeout -k 10 400 python3 tools/shard_rehearsal.py --world $w --log-gates $lg --steps 8 >> $out/rehearsal_2p$lg.jsonl 2> $out/rehearsal.err || { echo "world $w failed"; tail -3 $out/rehearsal.err; exit 1; }
    tail -1 $out/rehearsal_2p$lg.jsonl
  done
done
for lg in 20 22; do
  bash tools/tl_shard.sh ev_${tag}_s8_$lg 8 $lg || exit 1
  mv gpurun_out/tl_ev_${tag}_s8_$lg.txt $out/timeline_2p${lg}_rank0_of_8.txt
done
bash tools/tl.sh ev_${tag}_16 - --log-gates 16 && mv gpurun_out/tl_ev_${tag}_16.txt $out/timeline_2p16.txt
bash tools/tl.sh ev_${tag}_20 - && mv gpurun_out/tl_ev_${tag}_20.txt $out/timeline_2p20.txt
echo done
