"""CPU suite, part 1: the C++ oracle and the Python model against the committed golden vectors and the external anchors.

External anchors (not produced by this repository): the merlin transcript known-answer test, the BN254 moduli and roots of unity,
the compressed generator, and the RNG-independent values of the reference's example circuit (src/main.rs:29-39).
"""
import json
import os

import numpy as np
import pytest

from oracle import pymodel as pm

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: json.load(open(os.path.join(G, name)))
I = lambda h: int(h, 16)


def test_external_anchors(orc):
    a = load("anchors.json")
    # merlin 3.0 test vector ("test protocol" / "some label" / "some data" / "challenge")
    assert a["merlin_kat"] == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    assert orc.transcript_kat().hex() == a["merlin_kat"]
    t = pm.Transcript(b"test protocol"); t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == a["merlin_kat"]
    assert I(a["r"]) == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 == pm.R_MOD
    assert I(a["q"]) == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 == pm.Q_MOD
    assert I(a["root_2_28"]) == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert I(a["omega_4"]) == 21888242871839275217838484774961031246007050428528088939761107053157389710902
    assert I(a["omega_16"]) == 14940766826517323942636479241147756311199852622225275649687664389641784935947
    assert I(a["omega_2_20"]) == 17220337697351015657950521176323262483320249231368149235373741788599650842711
    assert pow(pm.root_of_unity(1 << 20), 1 << 20, pm.R_MOD) == 1 and pow(pm.root_of_unity(1 << 20), 1 << 19, pm.R_MOD) == pm.R_MOD - 1
    assert a["generator_compressed"] == "01" + "00" * 31
    assert orc.g1_compress(orc.g1_generator(), 0).hex() == a["generator_compressed"]
    assert I(a["empty_plonk_proof_beta"]) == orc.fr_to_ints(orc.empty_beta())[0]


def test_example_circuit_known_answers():
    """src/main.rs:29-39 -> columns, sigma, s_id = X, selector polynomials."""
    a = load("anchors.json")["example"]
    v, g = pm.example_circuit()
    inp = pm.prepare_inputs(v, g)
    assert inp["sigma"] == [10, 1, 3, 2, 4, 6, 5, 7, 9, 8, 0, 11] == a["sigma"]
    assert inp["witness_flat"] == [3, 4, 7, 7, 2, 14, 14, 5, 19, 19, 3, 57]
    assert inp["s_id"] == [0, 1]
    half = 10944121435919637611123202872628637544274182200208017171849102093287904247809
    assert inp["q_add"] == [half, 0, half] and inp["q_mul"] == [half, 0, half - 1]
    assert inp["s_sigma"][0] == 0 and inp["s_sigma"][15] == 9194099743896221122520674968723232129798849434167497806676243317980593308842
    assert [I(x) for x in a["s_sigma"]] == inp["s_sigma"]


def test_oracle_field_and_curve(orc):
    f = load("field_g1.json")
    xs, ys = [I(x) for x in f["a"]], [I(y) for y in f["b"]]
    for name, conv, back, op in (("fr", orc.fr_from_ints, orc.fr_to_ints, orc.fr_op), ("fq", orc.fq_from_ints, orc.fq_to_ints, orc.fq_op)):
        a, b = conv(xs), conv(ys)
        for o in ("add", "sub", "mul"):
            assert back(op(o, a, b)) == [I(x) for x in f[name][o]], (name, o)
        assert back(op("inv", a[:24])) == [I(x) for x in f[name]["inv"]]
    gen = orc.g1_generator()
    for case in f["g1_mul_gen"]:
        out, inf = orc.g1_mul(gen, 0, orc.fr_from_ints([I(case["k"])])[0])
        assert orc.point_to_ints(out, inf) == (I(case["p"][0]), I(case["p"][1]))


def test_oracle_ntt(orc):
    for case in load("ntt.json"):
        a = orc.fr_from_ints([I(x) for x in case["input"]])
        g = orc.fr_from_ints([5])[0]
        assert orc.fr_to_ints(orc.ntt(a)) == [I(x) for x in case["fwd"]]
        assert orc.fr_to_ints(orc.ntt(a, inverse=True)) == [I(x) for x in case["inv"]]
        assert orc.fr_to_ints(orc.ntt(a, coset=g)) == [I(x) for x in case["coset5_fwd"]]
        assert orc.fr_to_ints(orc.ntt(a, inverse=True, coset=g)) == [I(x) for x in case["coset5_inv"]]


def _srs(orc, m):
    g = orc.points_from_ints([(I(p[0]), I(p[1])) for p in m["srs_g"]])
    gg = orc.points_from_ints([(I(p[0]), I(p[1])) for p in m["srs_gg"]])
    return g, gg


def test_oracle_msm_and_kzg(orc):
    m = load("msm_kzg.json")
    g, gg = _srs(orc, m)
    one = orc.fr_from_ints([1])[0]
    assert np.array_equal(orc.srs_powers(orc.fr_from_ints([I(m["tau"])])[0], one, 0, len(g)), g)
    for case in m["cases"]:
        sc = orc.fr_from_ints([I(s) for s in case["scalars"]]) if case["scalars"] else np.zeros((0, 4), np.uint64)
        out, inf = orc.msm(sc, g[: len(sc)])
        want = None if case["result"] is None else (I(case["result"][0]), I(case["result"][1]))
        assert orc.point_to_ints(out, inf) == want
    k = m["kzg"]
    coeffs = orc.fr_from_ints([I(c) for c in k["coeffs"]]); blind = orc.fr_from_ints([I(b) for b in k["blind"]])
    zeta, chal = orc.fr_from_ints([I(k["zeta"]), I(k["chal"])])
    c, inf = orc.kzg_commit(g, gg, coeffs, blind)
    assert orc.point_to_ints(c, inf) == (I(k["commit"][0]), I(k["commit"][1]))
    w, winf, rv = orc.kzg_open(g, gg, coeffs, blind, zeta, chal)
    assert orc.point_to_ints(w, winf) == (I(k["open_w"][0]), I(k["open_w"][1]))
    assert orc.fr_to_ints(rv)[0] == I(k["random_v"])
    assert orc.fr_to_ints(orc.poly_eval(coeffs, zeta))[0] == I(k["eval"])


@pytest.mark.parametrize("idx", [0, 1])
def test_oracle_whole_proof(orc, idx):
    """BASELINE configs[0]: the reference's example circuit, proved on the CPU; plus a 2^3-gate synthetic circuit."""
    p = load("proofs.json")[idx]
    variables = [I(x) for x in p["variables"]]
    gates = [tuple(g) for g in p["gates"]]
    inp = pm.prepare_inputs(variables, gates)
    one = orc.fr_from_ints([1])[0]
    tau, gs = orc.fr_from_ints([I(p["tau"]), I(p["gscal"])])
    g = orc.srs_powers(tau, one, 0, inp["n4"] + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    polys = [orc.fr_from_ints(inp[k]) for k in ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma"]]
    rc, proof, ch, _ = orc.prove(g, gg, polys, orc.fr_from_ints(inp["witness_flat"]), np.array(inp["sigma"], dtype=np.uint64), inp["n"].bit_length() - 1,
                                 orc.fr_from_ints([I(b) for b in p["blinds"]]), orc.fr_from_ints([I(p["chal"])])[0])
    assert rc == 0
    assert orc.fr_to_ints(ch) == [I(p["beta"]), I(p["gamma"]), I(p["alpha"]), I(p["zeta"])]
    assert proof.hex() == p["proof_hex"] and len(proof) == 1170
    zv, zc = orc.grand_product(orc.fr_from_ints(inp["witness_flat"]), np.array(inp["sigma"], dtype=np.uint64), ch[0], ch[1])
    assert pm.trim(orc.fr_to_ints(zc)) == [I(x) for x in p["z_coeffs"]]


def test_oracle_all_cores_msm_is_the_same_proof(orc):
    """The OpenMP form of the oracle's MSM (used only for the all-cores CPU baseline) yields the bytes of the single-threaded proof."""
    from plonk_amd import synth
    variables, gates = synth.synthetic_circuit(11, seed=4)
    tb = synth.circuit_tables(variables, gates)
    n4 = 4 * tb["n"]
    tau, gs = synth.seeded_scalars(2, 7)
    g = orc.srs_powers(tau, synth.to_mont([1])[0], 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    x_poly = np.zeros((n4, 4), dtype=np.uint64); x_poly[1] = synth.to_mont([1])[0]
    dom = orc.ntt(x_poly)
    sig = np.concatenate([np.asarray(tb["sigma"], dtype=np.int64), np.arange(3 * tb["n"], n4, dtype=np.int64)])
    polys = [synth.trim(tb["a"]), synth.trim(tb["b"]), synth.trim(tb["c"]), synth.trim(orc.ntt(tb["q_add_vals"], inverse=True)),
             synth.trim(orc.ntt(tb["q_mul_vals"], inverse=True)), synth.to_mont([0, 1]), synth.trim(orc.ntt(dom[sig], inverse=True))]
    blinds = synth.seeded_scalars(27, 8); chal = synth.to_mont([0x1234567890ABCDEF1122334455667788])[0]
    rc1, one, ch1, _ = orc.prove(g, gg, polys, tb["witness_flat"], tb["sigma"], tb["log_n"], blinds, chal)
    assert orc.set_threads(4) == 1
    try:
        rc4, four, ch4, _ = orc.prove(g, gg, polys, tb["witness_flat"], tb["sigma"], tb["log_n"], blinds, chal)
    finally:
        assert orc.set_threads(1) == 4
    assert rc1 == 0 and rc4 == 0 and one == four and np.array_equal(ch1, ch4)


def test_oracle_rejects_bad_witness(orc):
    v, g = pm.synthetic_circuit(3, seed=3)
    inp = pm.prepare_inputs(v, g)
    a, b, c, qa, qm = (orc.fr_from_ints(col) for col in inp["cols"])
    w = orc.fr_from_ints(inp["witness_flat"]); sigma = np.array(inp["sigma"], dtype=np.uint64)
    beta, gamma = orc.fr_from_ints([123456789, 987654321])
    zv, _ = orc.grand_product(w, sigma, beta, gamma)
    assert orc.quotient(a, b, c, qa, qm, zv, sigma, beta, gamma) == 0
    c[2] = a[0]
    assert orc.quotient(a, b, c, qa, qm, zv, sigma, beta, gamma) == 1


def test_oracle_public_input_term(orc):
    """build_public_input_poly (src/circuit.rs:410-439) in both restatements, on the reference's own example circuit (src/main.rs:29-39)
    with variable 0 marked public as the commented-out line src/main.rs:40 would: value 3, first used in gate 0, where a = 3."""
    inp = pm.prepare_inputs(*pm.example_circuit())
    n = inp["n"]
    a_col, b_col, c_col, qa, qm = inp["cols"]
    beta, gamma, alpha = 1234567, 7654321, 424242
    z_vals, _ = pm.build_grand_product(inp["witness_flat"], inp["sigma"], n, beta, gamma)
    f = lambda x: orc.fr_from_ints([x])[0]
    cols = [orc.fr_from_ints(col) for col in inp["cols"]]
    sigma = np.array(inp["sigma"], dtype=np.uint64)
    zv = orc.fr_from_ints(z_vals[:n])
    # matching public value: the term vanishes, t = 0 as without public inputs
    assert pm.build_quotient(a_col, b_col, c_col, qa, qm, z_vals, inp["sigma"], n, beta, gamma, alpha, [0], [a_col[0]]) == []
    assert orc.quotient_pi(*cols, zv, sigma, f(beta), f(gamma), f(alpha), [0], orc.fr_from_ints([a_col[0]])) == 0
    # a public variable whose first use is not the row's first wire (value 4 is b of gate 0, a is 3): alpha * (3 - 4) at row 0
    assert pm.public_input_values(a_col, n, alpha, [0], [b_col[0]])[0] == alpha * (a_col[0] - b_col[0]) % pm.R_MOD
    with pytest.raises(pm.NotDivisible):
        pm.build_quotient(a_col, b_col, c_col, qa, qm, z_vals, inp["sigma"], n, beta, gamma, alpha, [0], [b_col[0]])
    assert orc.quotient_pi(*cols, zv, sigma, f(beta), f(gamma), f(alpha), [0], orc.fr_from_ints([b_col[0]])) == 1
    # assignment semantics: the later entry for a row wins
    assert orc.quotient_pi(*cols, zv, sigma, f(beta), f(gamma), f(alpha), [0, 0], orc.fr_from_ints([b_col[0], a_col[0]])) == 0
    assert orc.quotient_pi(*cols, zv, sigma, f(beta), f(gamma), f(alpha), [0, 0], orc.fr_from_ints([a_col[0], b_col[0]])) == 1
    assert orc.quotient_pi(*cols, zv, sigma, f(beta), f(gamma), f(alpha), [n], orc.fr_from_ints([1])) == 2


def test_oracle_extended_quotient_satisfies_its_identity(orc):
    """orc.quotient_coset (the checker of plk_quotient_coset) against Python integers: t(x) (x^n - 1) = num(x) at a point off the domain."""
    log_n = 3
    variables, gates = pm.synthetic_circuit(log_n, 9)
    inp = pm.prepare_inputs(variables, gates)
    n, R = inp["n"], pm.R_MOD
    rng = pm.SplitMix64(123)
    beta, gamma, alpha, x = (rng.next_fr() for _ in range(4))
    z_vals, _ = pm.build_grand_product(inp["witness_flat"], inp["sigma"], n, beta, gamma)
    cols = [orc.fr_from_ints(c) for c in inp["cols"]]
    f = lambda v: orc.fr_from_ints([v])[0]
    rc, t = orc.quotient_coset(*cols, orc.fr_from_ints(z_vals[:n]), np.array(inp["sigma"], dtype=np.uint64), f(beta), f(gamma), f(alpha))
    assert rc == 0
    t = orc.fr_to_ints(t)
    assert any(t[:3 * n - 3]) and not any(t[3 * n - 3:])
    w = pow(5, (R - 1) // n, R)
    horner = lambda cs, at: sum(c * pow(at, i, R) for i, c in enumerate(cs)) % R
    interp = lambda vals: pm.ntt(list(vals), n, inverse=True)
    a, b, c, qa, qm = (interp(col) for col in inp["cols"])
    zc = interp(z_vals[:n])
    A, B, Cc, QA, QM = (horner(p, x) for p in (a, b, c, qa, qm))
    lhs, rhs = horner(zc, x), horner(zc, w * x % R)
    for k, wv in enumerate((A, B, Cc)):
        lhs = lhs * (wv + beta * horner(interp([3 * i + k for i in range(n)]), x) + gamma) % R
        rhs = rhs * (wv + beta * horner(interp([inp["sigma"][3 * i + k] for i in range(n)]), x) + gamma) % R
    num = (QA * (A + B - Cc) + QM * (A * B - Cc) + alpha * (lhs - rhs)) % R
    assert horner(t, x) * (pow(x, n, R) - 1) % R == num
    bad = list(inp["cols"][2]); bad[1] = (bad[1] + 1) % R
    rc, _ = orc.quotient_coset(cols[0], cols[1], orc.fr_from_ints(bad), cols[3], cols[4], orc.fr_from_ints(z_vals[:n]),
                               np.array(inp["sigma"], dtype=np.uint64), f(beta), f(gamma), f(alpha))
    assert rc == 1


def test_public_inputs_enter_the_transcript(orc):
    """absorb_public_inputs (src/transcript.rs:48-52): one "public_inputs" message holding every element, absent when there are none.
    The C++ oracle and the Python model must agree with public inputs present, and the proof must differ from the one without."""
    variables, gates = pm.example_circuit()
    inp = pm.prepare_inputs(variables, gates)
    rng = pm.SplitMix64(77)
    tau, gs = rng.next_fr(), rng.next_fr()
    srs_g, srs_gg = pm.make_srs(inp["n4"] + 1, tau, gs)
    blinds = [rng.next_fr() for _ in range(27)]
    chal = rng.next_u64() | (rng.next_u64() << 64)
    pub = [57, pm.R_MOD - 1, 0]
    want = pm.prove(inp, srs_g, srs_gg, blinds, chal, public_inputs=pub)
    polys = [orc.fr_from_ints(inp[k]) for k in ["a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma"]]
    g = orc.points_from_ints(srs_g); gg = orc.points_from_ints(srs_gg)
    args = (g, gg, polys, orc.fr_from_ints(inp["witness_flat"]), np.array(inp["sigma"], dtype=np.uint64), 2, orc.fr_from_ints(blinds), orc.fr_from_ints([chal])[0])
    rc, proof, ch, _ = orc.prove(*args, public_inputs=orc.fr_from_ints(pub))
    assert rc == 0 and proof == want["bytes"]
    rc, without, _, _ = orc.prove(*args)
    assert rc == 0 and without != proof
    # the first round's commitments do not depend on the transcript; everything after beta does
    assert without[9 * 32:9 * 32 + 7 * 33] == proof[9 * 32:9 * 32 + 7 * 33]
    assert without[:32] != proof[:32]
