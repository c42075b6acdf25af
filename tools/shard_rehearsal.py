#!/usr/bin/env python3
"""Per-rank cost of a sharded proof without the other ranks: this process plays rank 0 of `--world` on one GPU (its share of the
SRS only) and the exchange of partial sums is replaced by a local copy, so the proof bytes are wrong but every kernel and host step
of one rank runs as it would on a node.  Shows what strong scaling can be expected before a multi-GPU node is available.

    python tools/shard_rehearsal.py --world 8 --log-gates 20
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonk_amd import synth
from plonk_amd.prover import ProverKey

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--log-gates", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--log-block", type=int, default=0, help="points per block of the cyclic shards, as a power of two (0: what bench.py uses at this size)")
args = ap.parse_args()
n = 1 << args.log_gates
variables, gates = synth.synthetic_circuit(args.log_gates, seed=1)
tables = synth.circuit_tables(variables, gates)
pk = ProverKey(0)
tau, gscal = synth.seeded_scalars(2, 7)
total = 4 * n + 1
log_block = args.log_block or synth.cyclic_log_block(args.log_gates)
pk.generate_srs_cyclic(tau, gscal, total, 0, args.world, log_block=log_block)
if args.world > 1:
    pk.set_sharding(0, args.world, lambda mine: np.broadcast_to(mine, (args.world,) + mine.shape))
pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
inputs = pk.prepare(tables)
blinds = synth.seeded_scalars(27, 8)
for _ in range(2):
    pk.prove(inputs, blinds)
pk.ctx.sync()
t0 = time.perf_counter(); stage = np.zeros(6)
for _ in range(args.steps):
    stage += pk.prove(inputs, blinds).seconds
pk.ctx.sync()
ms = (time.perf_counter() - t0) / args.steps * 1e3
print(json.dumps({"world": args.world, "log_gates": args.log_gates, "log_block": log_block, "ms_per_proof_one_rank": round(ms, 3),
                  "stage_ms": {k: round(float(v) / args.steps * 1e3, 3) for k, v in zip(["round1_commit", "grand_product", "quotient", "open", "assemble", "total"], stage)}}))
