#!/usr/bin/env python3
"""Two prover keys of one process on one GPU, each driven by its own host thread (the keys share the device's four streams, every
other resource is per key): time per proof with both in flight against one key alone.  Every proof is compared with the first one.

    python tools/two_in_flight.py --log-gates 20 --proofs 20
"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonk_amd import synth
from plonk_amd.prover import ProverKey

ap = argparse.ArgumentParser()
ap.add_argument("--log-gates", type=int, default=20)
ap.add_argument("--proofs", type=int, default=20)
args = ap.parse_args()
n = 1 << args.log_gates
tables = synth.circuit_tables(*synth.synthetic_circuit(args.log_gates, seed=1))
tau, gscal = synth.seeded_scalars(2, 7)
blinds = synth.seeded_scalars(27, 8)
keys = []
for _ in range(2):
    pk = ProverKey(0)
    pk.generate_srs(tau, gscal, 4 * n + 1)
    pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
    keys.append((pk, pk.prepare(tables)))
want = keys[0][0].prove(keys[0][1], blinds).proof_bytes
assert keys[1][0].prove(keys[1][1], blinds).proof_bytes == want

def run(pk, inputs, count, bad):
    for _ in range(count):
        if pk.prove(inputs, blinds).proof_bytes != want:
            bad.append(1)

bad = []
keys[0][0].ctx.sync()
t0 = time.perf_counter()
run(keys[0][0], keys[0][1], args.proofs, bad)
keys[0][0].ctx.sync()
solo = (time.perf_counter() - t0) / args.proofs * 1e3
threads = [threading.Thread(target=run, args=(pk, inputs, args.proofs, bad)) for pk, inputs in keys]
t0 = time.perf_counter()
for t in threads: t.start()
for t in threads: t.join()
for pk, _ in keys: pk.ctx.sync()
both = (time.perf_counter() - t0) / (2 * args.proofs) * 1e3
print(json.dumps({"log_gates": args.log_gates, "solo_ms_per_proof": round(solo, 3), "two_in_flight_ms_per_proof": round(both, 3), "differing_proofs": len(bad)}))
for pk, inputs in keys:
    inputs.close(); pk.close()
