#!/usr/bin/env python3
"""Benchmark of the prover hot path on MI355X: one step = one complete proof (src/prover.rs:119-336) of a synthetic
2^k-gate circuit whose inputs are already resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

With N > 1 every rank keeps every N-th block of 4096 SRS powers (and their window tables); each MSM is the sum of the N
partial sums, exchanged as 128-byte points over RCCL; the cheap non-MSM rounds are replicated.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAD_PEAK = 33.3e12          # v_mad_u64_u32 lane-ops/s measured on MI355X by tools/valu_microbench (profiles/valu_microbench_r01.txt)
HBM_PEAK = 8.0e12
MODMUL_PER_MADD = 10        # XYZZ mixed addition: 8 M + 2 S
MADS_PER_MODMUL = 136       # 8x8 + 8x8 + 8 multiply-adds of an 8x32-bit-limb Montgomery product (the algorithmic count)
ACC_ALGO_BYTES_PER_ENTRY = 68.0   # 64 B table point + 4 B sorted entry gathered per (pair, window)


def profile_derived():
    """Figures that come from rocprofv3 counter passes, not from this run: profiles/current.json, written by tools/profile_round.sh
    together with the hash of the kernel library it profiled.  Returned with stale = True when the library loaded now is another build."""
    import hashlib
    from plonk_amd import hip
    path = os.path.join(ROOT, "profiles", "current.json")
    try:
        with open(path) as f:
            cur = json.load(f)
        with open(hip.LIB_PATH, "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
    except (OSError, ValueError):
        return None
    cur["stale"] = bool(cur.get("library_sha16") != sha)
    return cur

def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pk, inputs, tables, blinds, chal, gpu_proof, log_gates, log_gates_sample, seed):
    """The CPU oracle (oracle/oracle.cpp: the restated reference CPU path, dead s_sigma_vals loop of src/prover.rs:240 and debug prints
    removed) on the host cores, as the checker being timed, never as part of the product path:
      all cores     one proof of the SAME 2^log_gates circuit the GPU just proved, MSMs spread over the host threads (OpenMP), measured
                    directly, and its 1170 bytes compared with the GPU's proof;
      single thread the reference's own setting (arkworks without the `parallel` feature, Cargo.toml:9-18) on a 2^log_gates_sample
                    circuit of the same family, scaled by the stated factor (the proof is MSM-dominated and linear in n).
    The SRS powers are read back from the device (the tests pin them against the oracle's own); they are inputs, not timed."""
    from oracle import oracle as orc
    from plonk_amd import synth
    n = tables["n"]
    n4 = 1 << (3 * n - 1).bit_length()
    g = pk.ctx.srs_read(0, 0, n4 + 1)
    gg = pk.ctx.srs_read(1, 0, 3)
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
    orc.set_threads(threads)
    t0 = time.perf_counter()
    rc, proof, _, stages_all = orc.prove(g, gg, inputs.polys, tables["witness_flat"], tables["sigma"], tables["log_n"], blinds, chal)
    dt_all = time.perf_counter() - t0
    orc.set_threads(1)
    assert rc == 0
    single = None
    if log_gates_sample:
        ls = min(log_gates_sample, log_gates)
        if ls == log_gates:
            tb, polys = tables, inputs.polys
        else:
            v2, g2 = synth.synthetic_circuit(ls, seed=seed)
            tb = synth.circuit_tables(v2, g2)
            ns = tb["n"]; n4s = 1 << (3 * ns - 1).bit_length()
            x_poly = np.zeros((n4s, 4), dtype=np.uint64)
            x_poly[1] = synth.to_mont([1])[0]
            domain = orc.ntt(x_poly)                                       # w^i, i < 4n
            sig = np.concatenate([np.asarray(tb["sigma"], dtype=np.int64), np.arange(3 * ns, n4s, dtype=np.int64)])
            polys = [synth.trim(tb["a"]), synth.trim(tb["b"]), synth.trim(tb["c"]), synth.trim(orc.ntt(tb["q_add_vals"], inverse=True)),
                     synth.trim(orc.ntt(tb["q_mul_vals"], inverse=True)), synth.to_mont([0, 1]), synth.trim(orc.ntt(domain[sig], inverse=True))]
        n4s = 1 << (3 * tb["n"] - 1).bit_length()
        t1 = time.perf_counter()
        rc1, _, _, stages = orc.prove(g[:n4s + 1], gg, polys, tb["witness_flat"], tb["sigma"], tb["log_n"], blinds, chal)
        dt = time.perf_counter() - t1
        assert rc1 == 0
        factor = 2 ** (log_gates - ls)
        single = {"value": round(dt * factor * 1e3, 1), "unit": "ms", "cores": 1, "measured_s": round(dt, 3), "sample_log_gates": ls, "scaled_by": factor,
                  "stage_s": [round(float(x), 3) for x in stages],
                  "sample": f"1 thread (the reference's arkworks feature set) proving a 2^{ls}-gate circuit of the same family in {dt:.2f} s"
                            + (f", scaled x{factor} to 2^{log_gates} gates" if factor > 1 else "")}
    return {
        "value": round(dt_all * 1e3, 1), "unit": "ms", "cores": threads, "kind": "port", "cpu_model": _cpu_model(),
        "sample": f"oracle/oracle.cpp (restated reference CPU path: dead s_sigma_vals loop of src/prover.rs:240 and debug prints removed) proving the same "
                  f"2^{log_gates}-gate circuit once, MSMs spread over {threads} host threads with OpenMP: {dt_all:.2f} s measured, no scaling",
        "measured_s": round(dt_all, 3), "stage_s": [round(float(x), 3) for x in stages_all],
        "proof_bytes_equal_gpu": bool(proof == gpu_proof),
        "single_thread": single,
    }


def ntt_stress(ctx, rank, world, local_rank, backend, log_n, reps, dist, torch, library_exchange=False):
    """Standalone coset NTT of 2^log_n points (BASELINE.json configs[4]) outside the timed region of the proof: on one GPU the
    library's transform, on N GPUs the four-step form with one all-to-all (plonk_amd/dist_ntt.py).  Inputs resident in HBM."""
    from plonk_amd import dist_ntt
    ops = dist_ntt.HipOps(ctx, local_rank, backend=backend, library_exchange=library_exchange)
    sn = dist_ntt.ShardedNtt(ops, rank, world, log_n, coset=5)
    m = (1 << log_n) // world
    rng = np.random.default_rng(3 + rank)
    raw = rng.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 60) - 1)                      # < 2^252 < r: valid residues
    buf, tmp = ops.alloc(m), ops.alloc(m)
    ops.load(buf, raw)
    keep = raw[:64].copy()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(fn, a, b):
        a2 = fn(a, b)                                          # warm-up: twiddle and coset tables, scratch buffers
        b2 = b if a2 is a else a
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            nxt = fn(a2, b2)
            b2 = b2 if nxt is a2 else a2
            a2 = nxt
        fence()
        dt = (time.perf_counter() - t0) / reps
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=ops.device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, a2, b2

    # round trip first (forward then inverse must return the input bit for bit), then timing
    ev = sn.forward(buf, tmp)
    back = sn.inverse(ev, buf if ev is tmp else tmp)
    ok = bool(np.array_equal(ops.read(back)[:64], keep))
    other = tmp if back is buf else buf
    fwd_s, a, b = timed(sn.forward, back, other)
    inv_s, a, b = timed(sn.inverse, a, b)
    n = 1 << log_n
    mads = (n // 2) * log_n * MADS_PER_MODMUL
    return {
        "log_n": log_n, "coset_generator": 5, "n_gpus": world, "round_trip_bit_exact": ok,
        "forward_ms": round(fwd_s * 1e3, 4), "inverse_ms": round(inv_s * 1e3, 4),
        "GBps": round(64.0 * n / fwd_s / 1e9, 1), "hbm_frac": round(64.0 * n / fwd_s / (HBM_PEAK * world), 4),
        "mad_frac": round(mads / fwd_s / (MAD_PEAK * world), 4),
        "note": "algorithmic bytes 64 N (each element read and written once), algorithmic work N/2 log2 N modular products of 136 multiply-adds; "
                "the transform is bound by the integer VALU rate, not by HBM" + ("" if world == 1 else
                "; four-step split, coefficients cyclic over ranks, evaluations in column blocks, " +
                ("one plk_ntt_sharded_dev call per transform: the exchange is grouped ncclSend / ncclRecv on the library's stream" if library_exchange
                 else "one all_to_all_single over torch.distributed between the library's two halves")),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-gates", type=int, default=20)
    ap.add_argument("--cpu-log-gates", type=int, default=18, help="size of the single-thread CPU-baseline sample (0 = skip; the all-cores figure is measured at full size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-incl-h2d", action="store_true", help="skip the host-arrays-in measurement (value_incl_h2d_ms)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ntt-log-n", type=int, default=24, help="size of the standalone coset NTT reported beside the proof (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL over xGMI, one rank per GPU) or gloo (rehearsal with several ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            gather_device = torch.device("cuda", local_rank)
        else:   # rehearsal on fewer GPUs than ranks: the partial sums travel through host memory
            dist.init_process_group(backend=args.dist_backend)
            local_rank = local_rank % max(1, torch.cuda.device_count())
            gather_device = torch.device("cpu")

    from plonk_amd import synth
    from plonk_amd.prover import ProverKey

    log_n = args.log_gates
    n = 1 << log_n
    n4 = 4 * n
    library_comm = False
    t_setup = time.perf_counter()
    variables, gates = synth.synthetic_circuit(log_n, seed=args.seed)
    tables = synth.circuit_tables(variables, gates)
    del variables, gates
    pk = ProverKey(local_rank)
    tau, gscal = synth.seeded_scalars(2, 7)
    total = n4 + 1
    # block-cyclic point ranges (blocks of 4096 powers dealt to the ranks in turn): every MSM of the proof, the n-coefficient ones
    # and the 4n-coefficient one alike, is split evenly; contiguous ranges of the 4n + 1 powers would leave the first ranks
    # with all of the n-coefficient polynomials
    log_block = synth.cyclic_log_block(log_n)
    pk.generate_srs_cyclic(tau, gscal, total, rank, world, log_block=log_block)
    if world > 1:
        bufs = {}

        def gather(mine):
            count = mine.shape[0]
            if count not in bufs:
                bufs[count] = (torch.empty(16 * count, dtype=torch.int64, device=gather_device),
                               torch.empty(16 * count * world, dtype=torch.int64, device=gather_device))
            send, recv = bufs[count]
            send.copy_(torch.from_numpy(mine.view(np.int64).reshape(-1)))
            dist.all_gather_into_tensor(recv, send)
            return recv.cpu().numpy().view(np.uint64).reshape(world, count, 16)
        collective = "torch.distributed all_gather_into_tensor (%s)" % args.dist_backend
        if args.dist_backend == "nccl":
            # the exchange inside the library: rank 0's RCCL id travels through torch.distributed (which stays for the rendezvous, the
            # barriers and the timing reduction), the partial sums then go through ncclAllGather without passing through Python
            ok, why = 1, ""
            ids = [None]
            try:
                if rank == 0:
                    ids[0] = pk.comm_unique_id()
            except Exception as e:
                why = str(e)
            dist.broadcast_object_list(ids, src=0)            # every rank takes part, whether or not rank 0 obtained an id
            try:
                if ids[0] is None:
                    raise RuntimeError(why or "rank 0 has no RCCL id")
                pk.comm_init(ids[0], rank, world)
            except Exception as e:
                ok, why = 0, str(e)
            # the ranks must agree: one rank on the library's exchange and another on the fallback would wait for each other for ever
            agreed = torch.tensor([ok], dtype=torch.int32, device=gather_device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 1:
                library_comm = True
                collective = "ncclAllGather inside libplonk_hip (RCCL over xGMI)"
            else:                        # fall back to the gather above, still RCCL
                if why:
                    sys.stderr.write("library communicator unavailable on rank %d (%s): partial sums go through torch.distributed\n" % (rank, why))
                pk.set_sharding(rank, world, gather)
        else:
            pk.set_sharding(rank, world, gather)
    pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
    inputs = pk.prepare(tables)
    blinds = synth.seeded_scalars(27, 8)
    setup_s = time.perf_counter() - t_setup

    def sync():
        pk.ctx.sync()
        if world > 1:
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        res = pk.prove(inputs, blinds)
    # Inside the timed region only the dominant kernel is bracketed with events (two per MSM); the per-class breakdown of all kernels
    # comes from two more proofs after it, outside the timed region (bracketing every class costs about 0.3 ms per proof).
    pk.ctx.profile_enable(2)
    pk.ctx.profile_reset()
    sync()
    t0 = time.perf_counter()
    stage = np.zeros(6)
    for _ in range(args.steps):
        res = pk.prove(inputs, blinds)
        stage += res.seconds
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=gather_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # The same proof with the reference's call shape (src/prover.rs:119-133 borrows host vectors): host arrays in, proof out, the
    # transfers (a, b, c, q_add, q_mul, s_id, s_sigma, witness_flat, sigma) inside the timed region.  Never `value`: reported beside it.
    incl = None
    if world == 1 and not args.no_incl_h2d:
        pk.ctx.profile_enable(0)
        host_arrays = [np.ascontiguousarray(p) for p in inputs.polys] + [np.ascontiguousarray(tables["witness_flat"], dtype=np.uint64),
                                                                         np.ascontiguousarray(tables["sigma"], dtype=np.uint64)]
        pinned = []
        for a in host_arrays:
            h = pk.ctx.host_empty(a.shape, a.dtype); h[...] = a; pinned.append(h)
        incl = {"bytes_per_proof": int(sum(a.nbytes for a in host_arrays))}
        for label, arrs, reps in (("pinned", pinned, args.steps), ("pageable", host_arrays, max(1, args.steps // 2))):
            r = pk.prove_host(arrs[:7], arrs[7], arrs[8], log_n, blinds)          # warm-up: device buffers of the key
            assert r.proof_bytes == res.proof_bytes, "proof from host arrays differs from the resident one"
            sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                pk.prove_host(arrs[:7], arrs[7], arrs[8], log_n, blinds)
            sync()
            incl[label + "_ms"] = round((time.perf_counter() - t1) / reps * 1e3, 3)
        for h in pinned:
            pk.ctx.host_free(h)
        pk.ctx.profile_enable(2)

    prof = {"msm_accumulate": pk.ctx.profile_get("msm_accumulate")}
    acc_busy_ms = pk.ctx.profile_get_busy("msm_accumulate")
    breakdown_steps = 2
    pk.ctx.profile_enable(1)
    pk.ctx.profile_reset()
    for _ in range(breakdown_steps):
        pk.prove(inputs, blinds)
    sync()
    breakdown = {name: pk.ctx.profile_get(name) for name in ("msm_accumulate", "msm_sort", "msm_reduce", "ntt", "pointwise", "division")}
    # dominant kernel: k_msm_accumulate.  One launch per MSM; per proof the launches cover 20n + 45 (point, scalar) pairs in total.
    acc_ms, acc_launches = prof["msm_accumulate"]
    pairs_per_proof = (20 * n + 45) / world
    launches_per_proof = max(1, acc_launches // max(1, args.steps))
    avg_launch_ms = acc_ms / max(1, acc_launches)
    pairs_per_launch = pairs_per_proof / launches_per_proof
    # Mixed additions per proof: every scalar of an MSM costs one per window of the plan the library uses at that length (15 windows
    # of 17 bits for long MSMs, 16 of 16 bits, 20 of 13 bits for short ones; plk_msm_plan_query).  The MSMs of a proof: commitments
    # to the seven circuit polynomials, z and t (three blinding terms each), then their nine openings (one coefficient fewer, two
    # hiding terms); t is the zero polynomial for a satisfied circuit.
    poly_lens = [len(p) for p in inputs.polys] + [n, 0]
    msm_lens = [(l, 3) for l in poly_lens] + [(max(l - 1, 0), 2) for l in poly_lens]
    entries_per_proof = sum((l + b) * pk.ctx.msm_plan(l)[1] for l, b in msm_lens) / world
    windows_avg = entries_per_proof / (sum(l + b for l, b in msm_lens) / world)
    entries_per_launch = entries_per_proof / launches_per_proof
    mads_per_entry = MODMUL_PER_MADD * MADS_PER_MODMUL
    mads_per_launch = entries_per_launch * mads_per_entry
    achieved = mads_per_launch / (avg_launch_ms * 1e-3) if avg_launch_ms > 0 else 0.0
    # Consecutive launches alternate between two streams and overlap, so a launch's own duration includes the time it shares the
    # chip with its neighbour.  "busy" is the time at least one launch was running; the isolated figure below times launches
    # of one 2^k-coefficient commitment with nothing else on the device (outside the timed region, not part of `value`).
    achieved_busy = entries_per_proof * args.steps * mads_per_entry / (acc_busy_ms * 1e-3) if acc_busy_ms > 0 else 0.0
    pk.ctx.profile_reset()
    iso_reps = 3
    iso_poly = np.ascontiguousarray(inputs.polys[0])
    d_iso = pk.ctx.upload(iso_poly)
    for _ in range(iso_reps):
        ticket = pk.ctx.commit_begin(d_iso, len(iso_poly), blinds[:3])
        if world > 1:
            pk.ctx.end_partial(ticket)
        else:
            pk.ctx.commit_end(ticket)
    pk.ctx.free(d_iso)
    iso_ms, iso_n = pk.ctx.profile_get("msm_accumulate")
    iso_launch_ms = iso_ms / max(1, iso_n)
    iso_pairs = len(iso_poly) // world + (3 if rank == 0 else 0)
    iso_windows = pk.ctx.msm_plan(len(iso_poly))[1]
    achieved_iso = iso_pairs * iso_windows * mads_per_entry / (iso_launch_ms * 1e-3) if iso_launch_ms > 0 else 0.0
    pk.ctx.profile_enable(False)
    bytes_per_launch = pairs_per_launch * 96.0
    derived = profile_derived()
    acc_prof = (derived or {}).get("k_msm_accumulate")
    traffic_per_launch = entries_per_launch * acc_prof["traffic_bytes_per_entry"] if acc_prof else None
    # The object leads with the figure that fits inside ms_per_step: algorithmic multiply-adds of the whole proof over the time at
    # least one accumulation kernel was running.  The per-launch figure (algorithmic work of one launch over its own duration, what
    # rocprofv3 reports as the kernel's average) follows as a sub-field: two launches overlap nearly all the time, so each sees half
    # a chip and its duration counts the shared time twice.
    roofline = {
        "kernel": "k_msm_accumulate", "bound": "valu_int", "achieved": round(achieved_busy / 1e12, 3), "peak": MAD_PEAK / 1e12,
        "unit": "T mad/s (algorithmic 32-bit multiply-adds: pairs x windows of the plan in use x 10 modmul x 136)", "windows_per_pair_avg": round(windows_avg, 3), "frac": round(achieved_busy / MAD_PEAK, 4),
        "basis": "busy time: all launches of a proof over the time at least one of them was running (HIP events on their streams)",
        "busy_ms_per_proof": round(acc_busy_ms / args.steps, 3),
        "traffic": round(traffic_per_launch) if traffic_per_launch is not None else None,
        "traffic_unit": "HBM-side bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE of this command, profiles/current.json); algorithmic: %d" % round(entries_per_launch * ACC_ALGO_BYTES_PER_ENTRY),
        "per_launch": {"note": "algorithmic work of one launch over its own duration; launches alternate between two streams and overlap",
                       "avg_launch_ms": round(avg_launch_ms, 4), "launches_per_proof": int(launches_per_proof), "pairs_per_launch_avg": int(pairs_per_launch),
                       "achieved": round(achieved / 1e12, 3), "frac": round(achieved / MAD_PEAK, 4)},
        "isolated": {"note": "one commitment of 2^%d coefficients at a time, outside the timed region" % log_n, "avg_launch_ms": round(iso_launch_ms, 4),
                     "pairs": int(iso_pairs), "windows": iso_windows, "achieved": round(achieved_iso / 1e12, 3), "frac": round(achieved_iso / MAD_PEAK, 4)},
        "hbm": {"bound": "hbm", "achieved": round(bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9, 1) if avg_launch_ms > 0 else 0.0, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": round(bytes_per_launch / (avg_launch_ms * 1e-3) / HBM_PEAK, 4) if avg_launch_ms > 0 else 0.0, "bytes_per_pair": 96,
                "traffic_GBps": round(traffic_per_launch / (avg_launch_ms * 1e-3) / 1e9, 1) if avg_launch_ms > 0 and traffic_per_launch is not None else None},
    }
    if acc_prof:
        roofline["instruction_stream"] = {
            "note": "rocprofv3 SQ counters of a standalone 2^20-point launch (profiles/current.json; not measured in this run): VALU wave-instructions per "
                    "mixed addition, the fraction of the roofline a stream of that length can reach at one instruction per four cycles (1360 "
                    "algorithmic multiply-adds / executed instructions), the share of the kernel's cycles in which the VALU pipes issue",
            "valu_instructions_per_addition": acc_prof["valu_instructions_per_addition"],
            "ceiling_frac": round(1360.0 / acc_prof["valu_instructions_per_addition"], 4), "valu_busy": acc_prof["valu_busy"]}
        roofline["profile"] = {"tag": derived.get("tag"), "library_sha16": derived.get("library_sha16"), "stale": derived["stale"]}
    else:
        roofline["profile"] = {"stale": True, "note": "profiles/current.json missing: traffic and instruction-stream figures not available"}
    out = {
        "metric": "prove time @2^%d gates" % log_n, "value": round(ms_per_step, 3), "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "synthetic 2^%d-gate random arithmetic circuit, BN254, hiding KZG10, full proof (BASELINE.json configs[2])" % log_n,
                   "log_gates": log_n, "srs_points": total, "msm_pairs_per_proof": 20 * n + 45, "msm_additions_per_proof": int(round(entries_per_proof * world)), "ntt_per_proof": 4,
                   "parallelism": "1 GPU" if world == 1 else "MSM sharded over %d GPUs by block-cyclic point ranges (%d points per block), partial sums exchanged by %s; other rounds replicated" % (world, 1 << log_block, collective)},
        "msm_gpts_per_s": round((20 * n + 45) / (ms_per_step * 1e-3) / 1e9, 4),
        "stage_ms": {k: round(float(v) / args.steps * 1e3, 3) for k, v in zip(["round1_commit", "grand_product", "quotient", "open", "assemble", "total"], stage)},
        "kernel_ms_per_proof": {k: round(v[0] / breakdown_steps, 3) for k, v in breakdown.items()},
        "roofline": roofline,
        "setup_s": round(setup_s, 2),
    }
    if incl:
        out["value_incl_h2d_ms"] = incl["pinned_ms"]
        out["incl_h2d"] = dict(incl, note="host arrays in, proof out (the call shape of src/prover.rs:119-133): the inputs cross PCIe while the first commitments "
                                          "run; pinned = page-locked host arrays (plk_host_malloc), pageable = ordinary numpy arrays, which the library copies through its ring of page-locked buffers with helper threads")
    if args.ntt_log_n:
        try:
            import torch as _torch
            out["ntt_stress"] = ntt_stress(pk.ctx, rank, world, local_rank, args.dist_backend, args.ntt_log_n, 5, dist, _torch, library_exchange=library_comm)
        except Exception as e:   # the proof line above stands on its own
            out["ntt_stress"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        # the last proof of the timed region through the pairing verifier (src/verifier.rs on the CPU, ~0.3 s, outside the timed region):
        # at N > 1 this is the proof the ranks assembled from their partial sums
        from plonk_amd import prover as _prover
        out["proof_verified"] = bool(_prover.verify(tau, gscal, res.proof_bytes, synth.to_mont([0x1234567890ABCDEF1122334455667788])[0]))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the CPU baseline is reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(pk, inputs, tables, blinds, synth.to_mont([0x1234567890ABCDEF1122334455667788])[0], res.proof_bytes, log_n,
                                           args.cpu_log_gates, args.seed)
    inputs.close()
    pk.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
