#!/usr/bin/env python3
"""Standalone MSM / NTT timing on one MI355X (BASELINE configs[4] style stress, sizes selectable).

    python tools/bench_msm.py --log-n 20 --reps 5 [--ntt]
Prints per-kernel-class device time (HIP events inside the library) and host wall time per call.
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonk_amd import synth
from plonk_amd.hip import Context

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ntt", action="store_true")
ap.add_argument("--quotient", action="store_true", help="time the extended-coset quotient of a 2^(log_n - 2)-gate circuit (4n = 2^log_n coset points) instead of the MSM")
ap.add_argument("--batch", type=int, default=0, help="also time this many hiding commitments in flight at once")
args = ap.parse_args()
n = 1 << args.log_n
ctx = Context(0)
if args.quotient:
    # synthetic columns: the kernels do the same work whatever the values are; status NOT_DIVISIBLE is expected and ignored
    from plonk_amd.hip import PlonkHipError
    gates = n // 4
    rng = np.random.default_rng(4)
    def col():
        raw = rng.integers(0, 1 << 63, size=(gates, 4), dtype=np.uint64); raw[:, 3] &= np.uint64((1 << 60) - 1); return ctx.upload(raw)
    cols = [col() for _ in range(6)]
    d_sigma = ctx.upload(rng.permutation(3 * gates).astype(np.uint64))
    d_t = ctx.malloc(n * 32)
    bga = synth.seeded_scalars(3, 9)
    def run():
        try:
            ctx.quotient_coset_dev(*cols, d_sigma, args.log_n - 2, bga[0], bga[1], bga[2], d_t)
        except PlonkHipError as e:
            assert e.status == "PLK_ERR_NOT_DIVISIBLE"
    run(); ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    wall = (time.perf_counter() - t0) / args.reps
    pw, pc = ctx.profile_get("pointwise"); nt, nc = ctx.profile_get("ntt")
    # k_quotient_coset: 11 reads + 1 write of 32 B per coset point
    print(json.dumps({"log_coset_points": args.log_n, "gates": gates, "wall_ms": round(wall * 1e3, 3), "pointwise_ms_per_call": round(pw / args.reps, 4),
                      "ntt_ms_per_call": round(nt / args.reps, 4), "ntt_launches_per_call": nc // args.reps,
                      "quotient_kernel_algorithmic_bytes": 12 * 32 * n}))
    sys.exit(0)
tau, gs = synth.seeded_scalars(2, 7)
t0 = time.perf_counter(); ctx.srs_generate(tau, gs, n); t_srs = time.perf_counter() - t0
rng = np.random.default_rng(2)
raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
raw[:, 3] &= np.uint64((1 << 60) - 1)          # < 2^252 < r: valid Montgomery residues of uniformly distributed values
d = ctx.upload(raw)
ctx.msm_dev(d, n)                               # warm-up (allocations)
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter()
for _ in range(args.reps):
    out = ctx.msm_dev(d, n)
wall = (time.perf_counter() - t0) / args.reps
res = {"log_n": args.log_n, "srs_setup_s": round(t_srs, 3), "msm_wall_ms": round(wall * 1e3, 3), "gpts_per_s": round(n / wall / 1e9, 4)}
for name in ("msm_sort", "msm_accumulate", "msm_reduce"):
    ms, cnt = ctx.profile_get(name)
    res[name + "_ms"] = round(ms / max(cnt, 1), 4)
if args.batch:
    blind = synth.seeded_scalars(3, 5)
    def round_of_commits():
        tickets = [ctx.commit_begin(d, n, blind) for _ in range(args.batch)]
        return [ctx.commit_end(t) for t in tickets]
    round_of_commits(); round_of_commits(); round_of_commits(); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        round_of_commits()
    w = (time.perf_counter() - t0) / args.reps
    res.update(batch=args.batch, batch_wall_ms=round(w * 1e3, 3), batch_ms_per_msm=round(w * 1e3 / args.batch, 3),
               batch_acc_ms=round(ctx.profile_get("msm_accumulate")[0] / max(1, ctx.profile_get("msm_accumulate")[1]), 4))
if args.ntt:
    g5 = synth.to_mont([5])[0]
    ctx.ntt_dev(d, args.log_n); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.ntt_dev(d, args.log_n)
    w = (time.perf_counter() - t0) / args.reps
    ms, cnt = ctx.profile_get("ntt")
    res.update(ntt_wall_ms=round(w * 1e3, 3), ntt_kernel_ms=round(ms / max(cnt, 1), 4), ntt_GBps=round(64 * n / (ms / max(cnt, 1) * 1e-3) / 1e9, 1))
    ctx.ntt_dev(d, args.log_n, coset=g5); ctx.ntt_dev(d, args.log_n, inverse=True, coset=g5)   # builds the coset factor tables
    for label, inv in (("coset_fwd", False), ("coset_inv", True)):
        ctx.profile_reset()
        for _ in range(args.reps):
            ctx.ntt_dev(d, args.log_n, inverse=inv, coset=g5)
        ms, cnt = ctx.profile_get("ntt")
        res[label + "_kernel_ms"] = round(ms / max(cnt, 1), 4)
        res[label + "_GBps"] = round(64 * n / (ms / max(cnt, 1) * 1e-3) / 1e9, 1)
print(json.dumps(res))
