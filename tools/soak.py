#!/usr/bin/env python3
"""Proves the same inputs many times and checks that every proof has the same bytes (a race between the library's streams would
show up as a differing proof sooner or later).    python tools/soak.py --log-gates 16 --proofs 300"""
import argparse, hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonk_amd import synth
from plonk_amd.prover import ProverKey

ap = argparse.ArgumentParser()
ap.add_argument("--log-gates", type=int, default=16)
ap.add_argument("--proofs", type=int, default=300)
args = ap.parse_args()
n = 1 << args.log_gates
tables = synth.circuit_tables(*synth.synthetic_circuit(args.log_gates, seed=1))
pk = ProverKey(0)
tau, gscal = synth.seeded_scalars(2, 7)
pk.generate_srs(tau, gscal, 4 * n + 1)
pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
inputs = pk.prepare(tables)
blinds = synth.seeded_scalars(27, 8)
first = pk.prove(inputs, blinds).proof_bytes
t0 = time.perf_counter()
bad = 0
for i in range(args.proofs):
    if pk.prove(inputs, blinds).proof_bytes != first:
        bad += 1
        print("proof", i, "differs", flush=True)
    if i % 50 == 49:
        print(i + 1, "proofs", round(time.perf_counter() - t0, 1), "s", flush=True)
print("log_gates", args.log_gates, "proofs", args.proofs, "differing", bad, "sha256", hashlib.sha256(first).hexdigest()[:16])
sys.exit(1 if bad else 0)
