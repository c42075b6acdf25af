#!/usr/bin/env python3
"""Writes the golden vectors under tests/golden/ from the big-integer model oracle/pymodel.py.

PARITY UNPINNED against the Rust reference: it has no tests or fixtures and cannot run here.  The vectors pin the C++ oracle
and the HIP library to the model; external anchors recorded in anchors.json are the published merlin known-answer test, the
BN254 constants and the RNG-independent values of the reference's example circuit (src/main.rs:29-39).
Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pymodel as pm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
H = lambda x: hex(x)


def pt(P):
    return None if P is None else [H(P[0]), H(P[1])]


def main():
    rng = pm.SplitMix64(2024)
    # ---- anchors
    t = pm.Transcript(b"test protocol"); t.append_message(b"some label", b"some data")
    v, g = pm.example_circuit()
    inp = pm.prepare_inputs(v, g)
    anchors = dict(
        merlin_kat=t.challenge_bytes(b"challenge", 32).hex(),
        empty_plonk_proof_beta=H(pm.Transcript(b"plonk_proof").challenge_scalar(b"beta")),
        r=H(pm.R_MOD), q=H(pm.Q_MOD), root_2_28=H(pm.ROOT_2_28), omega_4=H(pm.root_of_unity(4)), omega_16=H(pm.root_of_unity(16)),
        omega_2_20=H(pm.root_of_unity(1 << 20)), generator_compressed=pm.g1_compressed(pm.G1_GEN).hex(),
        example=dict(sigma=inp["sigma"], witness_flat=[H(x) for x in inp["witness_flat"]], s_id=[H(x) for x in inp["s_id"]],
                     s_sigma=[H(x) for x in inp["s_sigma"]], q_add=[H(x) for x in inp["q_add"]], q_mul=[H(x) for x in inp["q_mul"]]),
    )
    json.dump(anchors, open(os.path.join(OUT, "anchors.json"), "w"), indent=1)

    # ---- field / curve / transform vectors
    xs = [rng.next_fr() for _ in range(24)] + [0, 1, pm.R_MOD - 1, 2]
    ys = [rng.next_fr() for _ in range(24)] + [pm.R_MOD - 1, 0, pm.R_MOD - 1, pm.R_MOD - 2]
    field = dict(a=[H(x) for x in xs], b=[H(y) for y in ys])
    for name, p in (("fr", pm.R_MOD), ("fq", pm.Q_MOD)):
        field[name] = dict(add=[H((x + y) % p) for x, y in zip(xs, ys)], sub=[H((x - y) % p) for x, y in zip(xs, ys)],
                           mul=[H(x * y % p) for x, y in zip(xs, ys)], inv=[H(pm.finv(x, p)) for x in xs[:24]])
    ks = [1, 2, 3, rng.next_fr(), pm.R_MOD - 1]
    field["g1_mul_gen"] = [dict(k=H(k), p=pt(pm.g1_mul(k, pm.G1_GEN))) for k in ks]
    json.dump(field, open(os.path.join(OUT, "field_g1.json"), "w"), indent=1)

    ntt = []
    for log_n in (0, 1, 3, 5):
        a = [rng.next_fr() for _ in range(1 << log_n)]
        ntt.append(dict(log_n=log_n, input=[H(x) for x in a], fwd=[H(x) for x in pm.ntt(a)], inv=[H(x) for x in pm.ntt(a, inverse=True)],
                        coset5_fwd=[H(x) for x in pm.ntt(a, coset=5)], coset5_inv=[H(x) for x in pm.ntt(a, inverse=True, coset=5)]))
    json.dump(ntt, open(os.path.join(OUT, "ntt.json"), "w"), indent=1)

    tau, gscal = rng.next_fr(), rng.next_fr()
    srs_g, srs_gg = pm.make_srs(65, tau, gscal)
    msm = dict(tau=H(tau), gscal=H(gscal), srs_g=[pt(P) for P in srs_g], srs_gg=[pt(P) for P in srs_gg], cases=[])
    for n in (0, 1, 2, 7, 33):
        sc = [rng.next_fr() for _ in range(n)]
        if n >= 7:
            sc[1], sc[2], sc[3], sc[4] = 0, 1, pm.R_MOD - 1, 1 << 15
        msm["cases"].append(dict(scalars=[H(s) for s in sc], result=pt(pm.msm(sc, srs_g[:n]))))
    coeffs = [rng.next_fr() for _ in range(20)]
    blind = [rng.next_fr() for _ in range(3)]
    zeta = rng.next_fr(); chal = rng.next_u64() | (rng.next_u64() << 64)
    W, rv = pm.kzg_open(coeffs, srs_g, srs_gg, blind, zeta, chal)
    msm["kzg"] = dict(coeffs=[H(c) for c in coeffs], blind=[H(b) for b in blind], zeta=H(zeta), chal=H(chal),
                      commit=pt(pm.kzg_commit(coeffs, srs_g, srs_gg, blind)), open_w=pt(W), random_v=H(rv), eval=H(pm.poly_eval(coeffs, zeta)))
    json.dump(msm, open(os.path.join(OUT, "msm_kzg.json"), "w"), indent=1)

    # ---- whole proofs: the reference's example circuit (BASELINE configs[0]) and a 2^3-gate synthetic circuit
    proofs = []
    for name, (variables, gates), seed in (("example", pm.example_circuit(), 11), ("synthetic_2p3_seed3", pm.synthetic_circuit(3, seed=3), 12)):
        inp = pm.prepare_inputs(variables, gates)
        r2 = pm.SplitMix64(seed)
        blinds = [r2.next_fr() for _ in range(27)]
        chal = r2.next_u64() | (r2.next_u64() << 64)
        g_, gg_ = pm.make_srs(inp["n4"] + 1, tau, gscal)
        pr = pm.prove(inp, g_, gg_, blinds, chal)
        for k in pm.POLY_ORDER:
            Wk, rvk = pr["proofs"][k]
            assert pm.kzg_check_with_trapdoor(pr["comms"][k], Wk, pr["evals"][k], rvk, pr["zeta"], chal, tau, gscal)
        zv, zc = pm.build_grand_product(inp["witness_flat"], inp["sigma"], inp["n"], pr["beta"], pr["gamma"])
        proofs.append(dict(name=name, variables=[H(x) for x in variables], gates=gates, tau=H(tau), gscal=H(gscal), blinds=[H(b) for b in blinds], chal=H(chal),
                           beta=H(pr["beta"]), gamma=H(pr["gamma"]), alpha=H(pr["alpha"]), zeta=H(pr["zeta"]), z_coeffs=[H(x) for x in zc],
                           proof_hex=pr["bytes"].hex()))
    json.dump(proofs, open(os.path.join(OUT, "proofs.json"), "w"), indent=1)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
