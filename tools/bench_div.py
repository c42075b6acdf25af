#!/usr/bin/env python3
"""Device time of the opening's division (the three suffix-Horner kernels) for one polynomial at a time, nothing else on the chip.

    python tools/bench_div.py --log-n 20 --reps 10
"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonk_amd import synth
from plonk_amd.hip import Context

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
n = 1 << args.log_n
ctx = Context(0)
tau, gs = synth.seeded_scalars(2, 7)
ctx.srs_generate(tau, gs, n + 1)
rng = np.random.default_rng(2)
raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
raw[:, 3] &= np.uint64((1 << 60) - 1)
d = ctx.upload(raw)
blind = synth.seeded_scalars(3, 3); zeta, chal = synth.seeded_scalars(2, 4)
ctx.open_end(ctx.open_begin(d, n, blind, zeta, chal))      # warm-up
ctx.profile_enable(1); ctx.profile_reset()
for _ in range(args.reps):
    ctx.open_end(ctx.open_begin(d, n, blind, zeta, chal))
ms, launches = ctx.profile_get("division")
# algorithmic bytes: every coefficient read once and every quotient coefficient written once (64 B), 2 modular products each
print(json.dumps({"log_n": args.log_n, "division_ms": round(ms / args.reps, 4), "GBps": round(64.0 * n / (ms / args.reps * 1e-3) / 1e9, 1),
                  "hbm_frac": round(64.0 * n / (ms / args.reps * 1e-3) / 8e12, 4)}))
