#!/bin/bash
# Two independent prover processes sharing one GPU (each with its own context, SRS tables and streams): per-process time per proof
# while both run, against one process alone.  usage: tools/two_provers.sh [steps]
steps=${1:-20}
run() { timeout -k 5 300 python3 bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-incl-h2d --ntt-log-n 0 > gpurun_out/two_$1.json 2> gpurun_out/two_$1.err; }
run solo || { echo "solo run failed"; tail -3 gpurun_out/two_solo.err; exit 1; }
run a & pa=$!
run b & pb=$!
wait $pa || { echo "process a failed"; tail -3 gpurun_out/two_a.err; exit 1; }
wait $pb || { echo "process b failed"; tail -3 gpurun_out/two_b.err; exit 1; }
python3 - <<'PY'
import json
s = json.load(open('gpurun_out/two_solo.json'))['value']
a = json.load(open('gpurun_out/two_a.json'))['value']
b = json.load(open('gpurun_out/two_b.json'))['value']
print(json.dumps({"solo_ms_per_proof": s, "two_processes_ms_per_proof": [a, b], "amortized_ms_per_proof_if_fully_overlapped": round(1.0 / (1.0 / a + 1.0 / b), 3)}))
PY
