#!/bin/bash
# Copies what tools/profile_round.sh and tools/round_evidence.sh left under gpurun_out/ into profiles/<tag>_*; usage: tools/install_evidence.sh <tag>
tag=$1
rm -f profiles/${tag}_*
for f in bench20.json bench20_kernel_stats.txt bench20_pmc_FETCH_SIZE.txt bench20_pmc_WRITE_SIZE.txt msm20_pmc_sq.txt ntt24.json ntt24_kernel_stats.txt ntt24_pmc_FETCH_SIZE.txt ntt24_pmc_WRITE_SIZE.txt quotient22.json quotient22_kernel_stats.txt quotient22_pmc_FETCH_SIZE.txt quotient22_pmc_WRITE_SIZE.txt; do cp gpurun_out/prof_$tag/$f profiles/${tag}_$f; done
sed "s#gpurun_out/prof_$tag/#profiles/${tag}_#g" gpurun_out/prof_$tag/current.json > profiles/current.json
for f in bench16.json bench22.json rehearsal_2p20.jsonl rehearsal_2p22.jsonl timeline_2p16.txt timeline_2p20.txt timeline_2p20_rank0_of_8.txt timeline_2p22_rank0_of_8.txt; do cp gpurun_out/ev_$tag/$f profiles/${tag}_$f; done
cp gpurun_out/b_ev_${tag}_20.json profiles/${tag}_bench20_short.json
[ -f gpurun_out/bench_default_$tag.json ] && cp gpurun_out/bench_default_$tag.json profiles/${tag}_bench20_default.json
[ -f gpurun_out/pmcdiv_a.txt ] && cp gpurun_out/pmcdiv_a.txt profiles/${tag}_bench16_pmc_sq_side_kernels.txt
ls profiles | grep "^${tag}_" | wc -l
