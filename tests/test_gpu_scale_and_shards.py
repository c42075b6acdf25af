"""GPU suite, part 3: committed golden vectors through the HIP path, full-size proofs checked through the SRS trapdoor,
and point-range sharding (two logical shards on one device; two processes exchanging partial sums over gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_polys_equal, independent_polys
from oracle import pymodel as pm
from plonk_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
I = lambda h: int(h, 16)


def test_golden_vectors_through_the_c_abi(orc):
    from plonk_amd.hip import Context
    from plonk_amd.prover import ProverKey
    m = json.load(open(os.path.join(G, "msm_kzg.json")))
    g = orc.points_from_ints([(I(p[0]), I(p[1])) for p in m["srs_g"]]); gg = orc.points_from_ints([(I(p[0]), I(p[1])) for p in m["srs_gg"]])
    ctx = Context(0)
    ctx.srs_load(g, gg)
    for case in m["cases"]:
        sc = orc.fr_from_ints([I(s) for s in case["scalars"]]) if case["scalars"] else np.zeros((0, 4), np.uint64)
        out, inf = ctx.msm(sc)
        want = None if case["result"] is None else (I(case["result"][0]), I(case["result"][1]))
        assert orc.point_to_ints(out, inf) == want
    k = m["kzg"]
    coeffs = orc.fr_from_ints([I(c) for c in k["coeffs"]]); blind = orc.fr_from_ints([I(b) for b in k["blind"]])
    zeta, chal = orc.fr_from_ints([I(k["zeta"]), I(k["chal"])])
    c, inf = ctx.kzg_commit(coeffs, blind)
    assert orc.point_to_ints(c, inf) == (I(k["commit"][0]), I(k["commit"][1]))
    w, winf, rv, ev = ctx.kzg_open(coeffs, blind, zeta, chal)
    assert orc.point_to_ints(w, winf) == (I(k["open_w"][0]), I(k["open_w"][1]))
    assert orc.fr_to_ints(rv)[0] == I(k["random_v"]) and orc.fr_to_ints(ev)[0] == I(k["eval"])
    for case in json.load(open(os.path.join(G, "ntt.json"))):
        a = orc.fr_from_ints([I(x) for x in case["input"]]); g5 = orc.fr_from_ints([5])[0]
        assert orc.fr_to_ints(ctx.ntt(a)) == [I(x) for x in case["fwd"]]
        assert orc.fr_to_ints(ctx.ntt(a, inverse=True)) == [I(x) for x in case["inv"]]
        assert orc.fr_to_ints(ctx.ntt(a, coset=g5)) == [I(x) for x in case["coset5_fwd"]]
        assert orc.fr_to_ints(ctx.ntt(a, inverse=True, coset=g5)) == [I(x) for x in case["coset5_inv"]]
    ctx.close()
    pk = ProverKey(0)
    for p in json.load(open(os.path.join(G, "proofs.json"))):
        tables = synth.circuit_tables([I(x) for x in p["variables"]], [tuple(x) for x in p["gates"]])
        tau, gs = synth.to_mont([I(p["tau"]), I(p["gscal"])])
        pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
        pk.set_opening_challenge(synth.to_mont([I(p["chal"])])[0])
        inputs = pk.prepare(tables)
        res = pk.prove(inputs, synth.to_mont([I(b) for b in p["blinds"]]))
        assert res.proof_bytes.hex() == p["proof_hex"], p["name"]
        assert synth.from_mont(res.challenges) == [I(p["beta"]), I(p["gamma"]), I(p["alpha"]), I(p["zeta"])]
        inputs.close()
    pk.close()


def _decompress(b):
    q = pm.Q_MOD
    flags = b[31] & 0xC0
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    if flags & 0x40:
        return None
    y = pow((x * x * x + 3) % q, (q + 1) // 4, q)
    assert y * y % q == (x * x * x + 3) % q, "x is not on the curve"
    if (y > q - y) != bool(flags & 0x80):
        y = q - y
    return (x, y)


def _check_proof_with_trapdoor(orc, proof, zeta, polys, blinds, chal_int, tau_int, gs_int):
    """Every commitment equals f(tau) G + r(tau) gamma G, every evaluation equals f(zeta) and every opening satisfies
    (tau - zeta) W = c C - c v G - random_v gamma G.  polys: 9 coefficient arrays (Montgomery limbs) in proof order."""
    r = pm.R_MOD
    tau = synth.to_mont([tau_int])[0]
    zeta_int = synth.from_mont(zeta)[0]
    bl = synth.from_mont(blinds)
    for k in range(9):
        ev = int.from_bytes(proof[32 * k: 32 * k + 32], "little")
        C = _decompress(proof[288 + 33 * k: 288 + 33 * k + 32]); assert proof[288 + 33 * k + 32] == 0
        o = 288 + 297 + 65 * k
        W = _decompress(proof[o: o + 32]); assert proof[o + 32] == 1
        rv = int.from_bytes(proof[o + 33: o + 65], "little")
        f_tau = synth.from_mont(orc.poly_eval(polys[k], tau))[0] if len(polys[k]) else 0
        f_zeta = synth.from_mont(orc.poly_eval(polys[k], zeta))[0] if len(polys[k]) else 0
        b0, b1, b2 = bl[3 * k: 3 * k + 3]
        r_tau = (b0 + b1 * tau_int + b2 * tau_int * tau_int) % r
        r_zeta = (b0 + b1 * zeta_int + b2 * zeta_int * zeta_int) % r
        assert ev == f_zeta, f"evaluation {k}"
        assert C == pm.g1_mul((f_tau + r_tau * gs_int) % r, pm.G1_GEN), f"commitment {k}"
        assert rv == chal_int * r_zeta % r, f"random_v {k}"
        assert pm.kzg_check_with_trapdoor(C, W, ev, rv, zeta_int, chal_int, tau_int, gs_int), f"opening {k}"


@pytest.mark.parametrize("log_n", [16, 20, 22])
def test_full_size_proof_through_the_trapdoor(orc, log_n):
    """BASELINE configs[1], [2] and [3]: 2^16, 2^20 and 2^22 gates on one MI355X.  A proof is uniquely determined by its inputs, so the
    trapdoor identities below pin every commitment, evaluation and opening exactly without a 2^20-point MSM on the CPU."""
    from plonk_amd.prover import ProverKey
    tau_int, gs_int, chal_int = 0x1F2E3D4C5B6A79880123456789ABCDEF1F2E3D4C5B6A7988, 0x0A1B2C3D4E5F60718293A4B5C6D7E8F9, 0x1234567890ABCDEF1122334455667788
    pk = ProverKey(0)
    tables = synth.circuit_tables(*synth.synthetic_circuit(log_n, seed=1))
    n = tables["n"]
    tau, gs = synth.to_mont([tau_int, gs_int])
    pk.generate_srs(tau, gs, 4 * n + 1)
    pk.set_opening_challenge(synth.to_mont([chal_int])[0])
    blinds = synth.seeded_scalars(27, 8)
    inputs = pk.prepare(tables)
    res = pk.prove(inputs, blinds)
    _, z_coeffs = orc.grand_product(tables["witness_flat"], tables["sigma"], res.challenges[0], res.challenges[1])
    # the seven input polynomials the identities are checked with are interpolated by the oracle (4n-point transforms on the CPU),
    # and must equal what the device interpolated (src/poly_utils.rs:14-31, src/main.rs:190-191) byte for byte
    own = independent_polys(orc, tables)
    assert_polys_equal(inputs.polys, own)
    polys = own + [z_coeffs, np.zeros((0, 4), np.uint64)]
    _check_proof_with_trapdoor(orc, res.proof_bytes, res.challenges[3], polys, blinds, chal_int, tau_int, gs_int)
    if log_n == 16:   # the CPU oracle still proves 2^16 gates in seconds: compare every byte, with SRS powers computed on the CPU
        g = orc.srs_powers(tau, synth.to_mont([1])[0], 0, 4 * n + 1); gg = orc.srs_powers(tau, gs, 0, 3)
        assert np.array_equal(pk.ctx.srs_read(0, 0, 4 * n + 1), g) and np.array_equal(pk.ctx.srs_read(1, 0, 3), gg)
        rc, want, ch, _ = orc.prove(g, gg, own, tables["witness_flat"], tables["sigma"], log_n, blinds, synth.to_mont([chal_int])[0])
        assert rc == 0 and res.proof_bytes == want and np.array_equal(ch, res.challenges)
    inputs.close()
    pk.close()


def _random_residues(count, seed):
    """count uniformly random 252-bit values, read as Montgomery residues (every value below r is a valid one)."""
    raw = np.random.default_rng(seed).integers(0, 1 << 63, size=(count, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 60) - 1)
    return raw


@pytest.mark.parametrize("length", [(1 << 21) - 1, 1 << 21, (1 << 21) + 1, (1 << 21) + 8193, 3 * (1 << 20) + 5])
def test_division_around_the_change_of_block_size(orc, length):
    """Polynomials of 2^21 coefficients and more are divided in blocks of 8192 coefficients, shorter ones in blocks of 2048: the
    evaluation (the division's remainder) against the oracle's Horner pass, and the quotient through its commitment and the SRS
    trapdoor, (tau - zeta) W = c C - c v G - random_v gamma G, on both sides of the change."""
    from plonk_amd.hip import Context
    c = Context(0)
    tau, gs = synth.seeded_scalars(2, 51)
    c.srs_generate(tau, gs, length + 1)
    coeffs = _random_residues(length, 52)
    blind = synth.seeded_scalars(3, 53)
    zeta = synth.seeded_scalars(1, 54)[0]
    chal = synth.to_mont([0x1F2E3D4C5B6A79881122334455667788])[0]
    ev = orc.poly_eval(coeffs, zeta)
    d = c.upload(coeffs)
    assert np.array_equal(c.poly_eval_dev(d, length, zeta), ev)
    w, winf, rv, got_ev = c.open_end(c.open_begin(d, length, blind, zeta, chal))
    assert np.array_equal(got_ev, ev)
    comm, cinf = c.commit_end(c.commit_begin(d, length, blind))
    c.free(d)
    # through the trapdoor, all scalars as integers: C = (f(tau) + gs r(tau)) G and (tau - zeta) W = c C - c v G - random_v gs G
    r_mod = pm.R_MOD
    t, g_s, z, ch = (synth.from_mont(x.reshape(1, 4))[0] for x in (tau, gs, zeta, chal))
    f_tau = synth.from_mont(orc.poly_eval(coeffs, tau).reshape(1, 4))[0]
    v = synth.from_mont(ev.reshape(1, 4))[0]
    r = synth.from_mont(blind)
    r_at = lambda x: (r[0] + r[1] * x + r[2] * x * x) % r_mod
    C, W = orc.point_to_ints(comm, cinf), orc.point_to_ints(w, winf)
    assert C == pm.g1_mul((f_tau + g_s * r_at(t)) % r_mod, pm.G1_GEN)
    random_v = synth.from_mont(rv.reshape(1, 4))[0]
    assert random_v == ch * r_at(z) % r_mod
    assert pm.kzg_check_with_trapdoor(C, W, v, random_v, z, ch, t, g_s)
    c.close()


def test_grand_product_2p23_three_scan_levels(orc):
    """2^23 rows = 4096 scan blocks: the block products themselves need two blocks and a third level.  Random rows and a random
    permutation; z on H against the oracle."""
    log_n = 23
    n = 1 << log_n
    rng = np.random.default_rng(23)
    base = orc.fr_from_ints([int.from_bytes(rng.bytes(40), "little") for _ in range(4099)])
    w = base[rng.integers(0, len(base), 3 * n)]
    sigma = rng.permutation(3 * n).astype(np.uint64)
    beta, gamma = base[0], base[1]
    zv, _ = orc.grand_product(w, sigma, beta, gamma)
    from plonk_amd.hip import Context
    ctx = Context(0)
    d_w, d_s, d_v = ctx.upload(w), ctx.upload(sigma), ctx.malloc(n * 32)
    ctx.grand_product_dev_enqueue(d_w, d_s, log_n, beta, gamma, d_v, None)
    ctx.sync()
    got = ctx.download(d_v, (n, 4))
    for d in (d_w, d_s, d_v):
        ctx.free(d)
    ctx.close()
    assert np.array_equal(got, zv)


def test_stress_msm_2p24_through_the_trapdoor(orc):
    """BASELINE configs[4], MSM half: 2^24 SRS points [tau^i] G with uniformly random scalars (seed 2).  With the trapdoor the
    answer is (sum_i s_i tau^i) G: one Horner pass on the CPU instead of a 2^24-point MSM."""
    from plonk_amd.hip import Context
    tau_int, gs_int = 0x2B3C4D5E6F708192A3B4C5D6E7F8091A2B3C4D5E6F7081, 0x1122334455667788
    n = 1 << 24
    ctx = Context(0)
    tau, gs = synth.to_mont([tau_int, gs_int])
    ctx.srs_generate(tau, gs, n)
    sc = _random_residues(n, 2)
    d = ctx.upload(sc)
    out, inf = ctx.msm_dev(d, n)
    f_tau = synth.from_mont(orc.poly_eval(sc, tau))[0]
    assert orc.point_to_ints(out, inf) == pm.g1_mul(f_tau, pm.G1_GEN)
    # a window of the same scalars against the middle of the SRS: sum_i s_i tau^(off + i)
    k, off = (1 << 22) + 12345, (1 << 23) - 777
    out, inf = ctx.msm_dev(d, k, base_offset=off)
    want = synth.from_mont(orc.poly_eval(sc[:k], tau))[0] * pow(tau_int, off, pm.R_MOD) % pm.R_MOD
    assert orc.point_to_ints(out, inf) == pm.g1_mul(want, pm.G1_GEN)
    ctx.free(d)
    ctx.close()


def test_msm_lengths_around_the_plan_boundaries(orc):
    """Lengths at which the MSM's plan changes: the sort's workgroup size steps (1024 .. 8192 scalars per workgroup so that at least
    128 workgroups run), the 2^20-point ranges a long MSM is pipelined in (above 1.5 * 2^20), ragged tails of one scalar, and windows
    that start in the middle of the SRS.  Checked through the trapdoor: sum_i s_i tau^(off + i) G."""
    from plonk_amd.hip import Context
    tau_int, gs_int = 0x3C4D5E6F708192A3B4C5D6E7F8091A2B3C4D5E6F708192A3, 0x99AABBCCDDEEFF
    n = (1 << 22) + 5
    ctx = Context(0)
    tau, gs = synth.to_mont([tau_int, gs_int])
    ctx.srs_generate(tau, gs, n)
    sc = _random_residues(n, 12)
    d = ctx.upload(sc)
    cases = [(131071, 0), (131072, 1), (131073, 4097), (262145, 3), (524287, 0), (1048575, 9), (1048577, 0), (1572864, 0), (1572865, 2), (2097153, 0),
             (3145731, 11), (4194309, 0), (5, 4194304)]
    for k, off in cases:
        out, inf = ctx.msm_dev(d, k, base_offset=off)
        want = synth.from_mont(orc.poly_eval(sc[:k], tau))[0] * pow(tau_int, off, pm.R_MOD) % pm.R_MOD
        assert orc.point_to_ints(out, inf) == pm.g1_mul(want, pm.G1_GEN), (k, off)
    # the same through tickets, several in flight at once, with blinding terms over the gamma powers
    blind = synth.seeded_scalars(3, 5)
    bsum = sum(b * pow(tau_int, j, pm.R_MOD) for j, b in enumerate(synth.from_mont(blind))) * gs_int % pm.R_MOD
    lens = [1572865, 131073, 7, 1048577]
    tickets = [ctx.commit_begin(d, k, blind) for k in lens]
    for k, t in zip(lens, tickets):
        out, inf = ctx.commit_end(t)
        want = (synth.from_mont(orc.poly_eval(sc[:k], tau))[0] + bsum) % pm.R_MOD
        assert orc.point_to_ints(out, inf) == pm.g1_mul(want, pm.G1_GEN), k
    ctx.free(d)
    ctx.close()


@pytest.mark.parametrize("short_max,mid_max", [("0", "0"), ("0", "1000000"), ("1000000", "0")])
def test_structured_scalars_through_the_trapdoor(orc, monkeypatch, short_max, mid_max):
    """Scalars a real proof can produce and a uniform sample never does: all equal (every entry of a window lands in one bucket),
    all zero, +-1, small values, a single non-zero coefficient, the largest digits; on each of the three window plans (17-, 16-
    and 13-bit windows, forced through the environment for the whole context)."""
    from plonk_amd.hip import Context
    monkeypatch.setenv("PLK_MSM_SHORT_MAX", short_max)
    monkeypatch.setenv("PLK_MSM_MID_MAX", mid_max)
    tau_int, gs_int = 0x1F2E3D4C5B6A79880123456789ABCDEF, 0x0A1B2C3D
    n = 1 << 16
    r = pm.R_MOD
    ctx = Context(0)
    tau, gs = synth.to_mont([tau_int, gs_int])
    ctx.srs_generate(tau, gs, n)
    rng = pm.SplitMix64(77)
    one_value = rng.next_fr()
    cases = {
        "all equal": [one_value] * n,
        "all zero": [0] * n,
        "plus and minus one": [1 if i % 3 else r - 1 for i in range(n)],
        "small": [i % 97 for i in range(n)],
        "single coefficient": [0] * (n - 1) + [one_value],
        "extreme digits": [int(("8000", "8001", "7fff", "ffff")[i & 3] * 15, 16) for i in range(n)],   # +2^15, -(2^15 - 1), carries through every 16-bit window
        "extreme digits, 17 bits": [sum(((1 << 16), (1 << 16) + 1, (1 << 16) - 1, (1 << 17) - 1)[i & 3] << (17 * j) for j in range(14)) for i in range(n)],
        "extreme digits, 13 bits": [sum(((1 << 12), (1 << 12) + 1, (1 << 12) - 1, (1 << 13) - 1)[i & 3] << (13 * j) for j in range(19)) for i in range(n)],
        "two buckets": [0x00010001 * (1 + (i & 1)) for i in range(n)],
    }
    for name, vals in cases.items():
        sc = synth.to_mont(vals)
        out, inf = ctx.msm(sc)
        want = synth.from_mont(orc.poly_eval(sc, tau))[0]
        assert orc.point_to_ints(out, inf) == (pm.g1_mul(want, pm.G1_GEN) if want else None), name
    ctx.close()


@pytest.mark.parametrize("log_n", [24, 25, 17])
def test_stress_ntt_2p24_coset_round_trip(orc, log_n):
    """BASELINE configs[4], NTT half: 2^24 uniformly random elements (seed 3), forward coset NTT with g = 5, then the inverse.
    Outputs are spot-checked against Horner evaluations f(5 w^k); the round trip must give back every input bit.
    2^25 runs four passes instead of three, 2^17 mixes pass radices 2^6 and 2^5."""
    from plonk_amd.hip import Context
    n = 1 << log_n
    r = pm.R_MOD
    ctx = Context(0)
    a = _random_residues(n, 3)
    d = ctx.upload(a)
    g5 = synth.to_mont([5])[0]
    ctx.ntt_dev(d, log_n, coset=g5)
    vals = ctx.download(d, (n, 4))
    w = pow(5, (r - 1) // n, r)
    for k in (0, 1, 0xABCDEF % n, n - 1):
        x = synth.to_mont([5 * pow(w, k, r) % r])[0]
        assert np.array_equal(vals[k], orc.poly_eval(a, x)), k
    if log_n == 24:   # every one of the 2^24 outputs against the oracle's transform (the s_sigma size of a 2^22-gate proof)
        assert np.array_equal(vals, orc.ntt(a, coset=g5))
    ctx.ntt_dev(d, log_n, inverse=True, coset=g5)
    back = ctx.download(d, (n, 4))
    assert np.array_equal(back, a)
    ctx.free(d)
    ctx.close()


def test_extended_quotient_2p18_identity(orc):
    """plk_quotient_coset at 2^18 gates (4n = 2^20 coset points), where the oracle's division takes too long: the quotient must
    satisfy t(x) (x^n - 1) = num(x) at a random point.  Every evaluation on the right comes from an inverse NTT and a Horner
    sweep (other kernels than the coset path); the final arithmetic is done with Python integers."""
    import time
    from plonk_amd.hip import Context
    log_n = 18
    n, R = 1 << log_n, pm.R_MOD
    variables, gates = synth.synthetic_circuit(log_n, seed=5)
    tb = synth.circuit_tables(variables, gates)
    ctx = Context(0)
    beta, gamma, alpha, xm = synth.seeded_scalars(4, 77)
    bi, gi, ai, x = synth.from_mont(np.stack([beta, gamma, alpha, xm]))
    zv, _ = orc.grand_product(tb["witness_flat"], tb["sigma"], beta, gamma)
    d = {k: ctx.upload(v) for k, v in dict(a=tb["a"], b=tb["b"], c=tb["c"], qa=tb["q_add_vals"], qm=tb["q_mul_vals"], z=zv).items()}
    d_sigma = ctx.upload(tb["sigma"])
    d_t = ctx.malloc(4 * n * 32)
    ctx.quotient_coset_dev(d["a"], d["b"], d["c"], d["qa"], d["qm"], d["z"], d_sigma, log_n, beta, gamma, alpha, d_t)   # warm-up: tables
    t0 = time.perf_counter()
    ctx.quotient_coset_dev(d["a"], d["b"], d["c"], d["qa"], d["qm"], d["z"], d_sigma, log_n, beta, gamma, alpha, d_t)
    print("extended quotient, 2^18 gates: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    t = ctx.download(d_t, (4 * n, 4))
    assert not t[3 * n - 3:].any() and t[:3 * n - 3].any()
    w = pow(5, (R - 1) // n, R)

    def at(dptr, point):                                   # interpolant of the n values at dptr, evaluated at `point`
        ctx.ntt_dev(dptr, log_n, inverse=True)
        v = synth.from_mont(ctx.poly_eval_dev(dptr, n, synth.to_mont([point])[0]))[0]
        ctx.ntt_dev(dptr, log_n)
        return v
    A, B, C_, QA, QM = (at(d[k], x) for k in ("a", "b", "c", "qa", "qm"))
    lhs, rhs = at(d["z"], x), at(d["z"], w * x % R)
    sig = tb["sigma"].astype(np.int64).reshape(n, 3)
    for k, wv in enumerate((A, B, C_)):
        d_id = ctx.upload(synth.to_mont((3 * np.arange(n) + k).tolist()))
        d_sg = ctx.upload(synth.to_mont(sig[:, k].tolist()))
        lhs = lhs * (wv + bi * at(d_id, x) + gi) % R
        rhs = rhs * (wv + bi * at(d_sg, x) + gi) % R
        ctx.free(d_id); ctx.free(d_sg)
    num = (QA * (A + B - C_) + QM * (A * B - C_) + ai * (lhs - rhs)) % R
    tx = synth.from_mont(ctx.poly_eval_dev(d_t, 4 * n, xm))[0]
    assert tx * (pow(x, n, R) - 1) % R == num
    for p in list(d.values()) + [d_sigma, d_t]:
        ctx.free(p)
    ctx.close()


def test_two_logical_shards_on_one_device(orc):
    """Point-range sharding without a second GPU: two contexts each hold half of the SRS; partial sums add up to the full MSM."""
    from plonk_amd.hip import Context
    n = 5000
    tau, gs = synth.seeded_scalars(2, 31)
    full = Context(0); full.srs_generate(tau, gs, n)
    lo = Context(0); lo.srs_generate(tau, gs, n, shard_first=0, shard_len=2048)
    hi = Context(0); hi.srs_generate(tau, gs, n, shard_first=2048, shard_len=n - 2048)
    assert np.array_equal(hi.srs_read(0, 2048, 16), full.srs_read(0, 2048, 16))
    sc = synth.seeded_scalars(n, 32)
    for (k, off) in ((n, 0), (3000, 1000), (100, 10), (100, 3000)):
        want = full.msm(sc[:k], base_offset=off)
        parts = []
        for c in (lo, hi):
            d = c.upload(sc[:k])
            parts.append(c.msm_partial_dev(d, k, off))
            c.free(d)
        got = full.sum_partials(np.stack(parts))
        assert got[1] == want[1] and np.array_equal(got[0], want[0])
    for c in (full, lo, hi):
        c.close()


@pytest.mark.parametrize("world,log_block", [(2, 6), (4, 4), (8, 3), (3, 5), (8, 12)])
def test_block_cyclic_shards_on_one_device(orc, world, log_block):
    """Block-cyclic point ranges: `world` contexts hold every world-th block; their partial sums add up to the full MSM for any
    range, and loading the powers from the host gives the same shard as generating them."""
    from plonk_amd.hip import Context
    n = 5000
    tau, gs = synth.seeded_scalars(2, 31)
    full = Context(0); full.srs_generate(tau, gs, n)
    g_all = full.srs_read(0, 0, n); gg = full.srs_read(1, 0, 3)
    shards = []
    for r in range(world):
        c = Context(0)
        if r % 2 == 0:
            c.srs_generate_cyclic(tau, gs, n, r, world, log_block)
        else:
            c.srs_load_cyclic(g_all, gg if r == 0 else None, r, world, log_block)
        shards.append(c)
    B = 1 << log_block
    for r, c in enumerate(shards):                    # the first block of every rank that has one
        if r * B < n:
            cnt = min(B, n - r * B, 16)
            assert np.array_equal(c.srs_read(0, r * B, cnt), g_all[r * B: r * B + cnt])
    sc = synth.seeded_scalars(n, 32)
    for (k, off) in ((n, 0), (3000, 1000), (100, 10), (1, 4999), (0, 7), (B + 3, B - 1)):
        if off + k > n:
            continue
        want = full.msm(sc[:k], base_offset=off)
        parts = []
        for c in shards:
            d = c.upload(sc[:k])
            parts.append(c.msm_partial_dev(d, k, off))
            c.free(d)
        got = full.sum_partials(np.stack(parts))
        assert got[1] == want[1] and np.array_equal(got[0], want[0]), (k, off)
    # hiding commitment and opening: the gamma terms live on rank 0 only
    coeffs = sc[:777]; blind = synth.seeded_scalars(3, 33); zeta, chal = synth.seeded_scalars(2, 34)
    want_c = full.kzg_commit(coeffs, blind)
    want_o = full.kzg_open(coeffs, blind, zeta, chal)
    parts_c, parts_o = [], []
    for c in shards:
        d = c.upload(coeffs)
        parts_c.append(c.end_partial(c.commit_begin(d, len(coeffs), blind))[0])
        po, rv, ev = c.end_partial(c.open_begin(d, len(coeffs), blind, zeta, chal))
        parts_o.append(po)
        assert np.array_equal(rv, want_o[2]) and np.array_equal(ev, want_o[3])
        c.free(d)
    assert np.array_equal(full.sum_partials(np.stack(parts_c))[0], want_c[0])
    assert np.array_equal(full.sum_partials(np.stack(parts_o))[0], want_o[0])
    for c in [full] + shards:
        c.close()


WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from plonk_amd import synth
from plonk_amd.prover import ProverKey
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
log_gates, log_block = int(sys.argv[4]), int(sys.argv[5])
tables = synth.circuit_tables(*synth.synthetic_circuit(log_gates, seed=4))
total = 4 * tables["n"] + 1
per = (total + world - 1) // world
pk = ProverKey(0)
tau, gs = synth.seeded_scalars(2, 7)
if sys.argv[3] == "cyclic":
    pk.generate_srs_cyclic(tau, gs, total, rank, world, log_block=log_block)
else:
    pk.generate_srs(tau, gs, total, shard_first=rank * per, shard_len=min(per, total - rank * per))
def gather(mine):
    recv = [torch.zeros(mine.shape, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(recv, torch.from_numpy(mine.view(np.int64)))
    return np.stack([r.numpy().view(np.uint64) for r in recv])
pk.set_sharding(rank, world, gather)
pk.set_opening_challenge(synth.to_mont([0xABCDEF0123456789])[0])
inputs = pk.prepare(tables)
res = pk.prove(inputs, synth.seeded_scalars(27, 8))
open(sys.argv[2] + "/proof%d.bin" % rank, "wb").write(res.proof_bytes)
inputs.close(); pk.close()
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("layout,world,log_gates,log_block", [("contiguous", 2, 8, 5), ("cyclic", 2, 8, 5), ("cyclic", 4, 14, 12)])
def test_sharded_proof_several_processes(orc, tmp_path, layout, world, log_gates, log_block):
    """Several ranks (gloo, all on this box's one GPU): each holds its share of the SRS (one contiguous range, or every world-th
    block; the last case uses the block size and rank count of a real 4-GPU run on a 2^14-gate circuit), partial sums are
    all-gathered, and every rank finishes with the proof the unsharded prover and the CPU oracle produce."""
    from plonk_amd.prover import ProverKey
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), layout, str(log_gates), str(log_block)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=280)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    proofs = [open(tmp_path / ("proof%d.bin" % r), "rb").read() for r in range(world)]
    assert all(p == proofs[0] for p in proofs)
    tables = synth.circuit_tables(*synth.synthetic_circuit(log_gates, seed=4))
    pk = ProverKey(0)
    tau, gs = synth.seeded_scalars(2, 7)
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    chal = synth.to_mont([0xABCDEF0123456789])[0]
    pk.set_opening_challenge(chal)
    inputs = pk.prepare(tables)
    single = pk.prove(inputs, synth.seeded_scalars(27, 8)).proof_bytes
    # the oracle proves from inputs made on the CPU alone (its own SRS powers and interpolations), compared with the device's first
    polys = independent_polys(orc, tables)
    assert_polys_equal(inputs.polys, polys)
    n4 = 4 * tables["n"]
    g = orc.srs_powers(tau, synth.to_mont([1])[0], 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    assert np.array_equal(pk.ctx.srs_read(0, 0, n4 + 1), g) and np.array_equal(pk.ctx.srs_read(1, 0, 3), gg)
    rc, want, _, _ = orc.prove(g, gg, polys, tables["witness_flat"], tables["sigma"], log_gates, synth.seeded_scalars(27, 8), chal)
    assert rc == 0 and single == want == proofs[0]
    inputs.close(); pk.close()


def test_library_rccl_exchange_with_one_rank(orc):
    """plk_comm_* (ncclCommInitRank / ncclAllGather inside the library) executed with world = 1, the most one GPU allows: the
    communicator comes up on the context's device, an all-gather returns the partial sums it was given, the sum of the gathered
    partials is the commitment, and a key set up with plonk_pk_comm_init still proves.  More ranks need more GPUs: the combine logic
    above the exchange is covered by the logical-shard and gloo tests."""
    from plonk_amd.prover import ProverKey
    pk = ProverKey(0)
    uid = pk.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    pk.comm_init(uid, 0, 1)
    ctx = pk.ctx
    tau, gs = synth.seeded_scalars(2, 7)
    pk.generate_srs(tau, gs, 4 * 256 + 1)
    rng = np.random.default_rng(5)
    mine = rng.integers(0, 1 << 63, size=(9, 16), dtype=np.uint64)
    for count in (1, 2, 7, 9):
        assert np.array_equal(ctx.comm_all_gather_partials(mine[:count], 1), mine[:count].reshape(1, count, 16))
    coeffs = synth.seeded_scalars(300, 11); blind = synth.seeded_scalars(3, 12)
    d = ctx.upload(coeffs)
    t = ctx.commit_begin(d, len(coeffs), blind)
    part = ctx.end_partial(t)
    part = part[0] if isinstance(part, tuple) else part
    got = ctx.comm_all_gather_partials(np.asarray(part, dtype=np.uint64).reshape(1, 16), 1)
    xy, inf = ctx.sum_partials(got.reshape(-1, 16))
    want, winf = ctx.kzg_commit(coeffs, blind)
    assert inf == winf and np.array_equal(xy, want)
    ctx.free(d)
    # the key still proves (world = 1 takes the direct path), and a second communicator on the same context is refused
    variables, gates = synth.synthetic_circuit(6, seed=2)
    tables = synth.circuit_tables(variables, gates)
    pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
    inputs = pk.prepare(tables)
    blinds = synth.seeded_scalars(27, 8)
    a = pk.prove(inputs, blinds)
    from plonk_amd.hip import PlonkHipError
    with pytest.raises(PlonkHipError):
        pk.comm_init(uid, 0, 1)
    b = pk.prove(inputs, blinds)
    assert a.proof_bytes == b.proof_bytes
    inputs.close()
    pk.close()
