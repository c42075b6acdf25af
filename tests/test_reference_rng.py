"""The reference driver's randomness (ark_std::test_rng, Fr::rand, G1/G2::rand, KZG10::setup, test_sponge) restated twice:
oracle/refrng.py (Python integers) and plonk_amd/host/setup.cpp (the host library).  External pins: RFC 8439 2.3.2 for the ChaCha
core and rand 0.8's own value-stability answer for StdRng with the seed ark-std uses.  Beyond those two the reference's crates cannot
be run here, so the remaining checks are cross-checks of the two restatements plus algebraic properties: parity unpinned."""
import ctypes as C
import json
import os

import numpy as np

from oracle import refrng as R
from plonk_amd import prover, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "refrng.json")))


def fq_ints(words):
    """x * 2^256 mod q words -> integers"""
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 4)
    return [int.from_bytes(r.tobytes(), "little") * R.MONT_INV_Q % R.Q_MOD for r in w]


def test_chacha_core_matches_rfc8439():
    k = GOLD["rfc8439_block"]
    key = bytes.fromhex(k["key"])
    for block in (R.chacha_block(key, k["counter"], k["stream"], k["rounds"]), [int(x) for x in prover.chacha_block(key, k["counter"], k["stream"], k["rounds"])]):
        assert block[0] == int(k["first_word"], 0) and block[15] == int(k["last_word"], 0)


def test_test_rng_matches_rand_value_stability_answer():
    """rand 0.8 rngs/std.rs: StdRng::from_seed([1,0,0,0, 23,0,0,0, 200,1,0,0, 210,30,0,0, 0...]).next_u64()"""
    assert R.test_rng().next_u64() == GOLD["rand08_first_u64"]
    assert prover.StdRng().next_u64() == GOLD["rand08_first_u64"]
    assert prover.StdRng(R.TEST_RNG_SEED).next_u64() == GOLD["rand08_first_u64"]


def test_block_buffer_word_order():
    """BlockRng: u32 and u64 reads interleaved, including a u64 that straddles two 64-word buffers"""
    a, b = prover.StdRng(), R.test_rng()
    assert [a.next_u64() for _ in range(40)] == [int(x, 0) for x in GOLD["next_u64"]]
    a = prover.StdRng()
    for i in range(400):
        if i % 7 == 3:
            assert a.next_u32() == b.next_u32()
        else:
            assert a.next_u64() == b.next_u64()
    # 63 single words, then a u64: low half = last word of this buffer, high half = first word of the next
    a, b = prover.StdRng(), R.test_rng()
    first = [a.next_u32() for _ in range(63)]
    assert first == [b.next_u32() for _ in range(63)]
    ref = R.test_rng()
    ref._refill(0)
    last_word = ref.buf[63]
    ref._refill(0)
    assert a.next_u64() == b.next_u64() == last_word | (ref.buf[0] << 32)
    assert a.next_u32() == ref.buf[1]


def test_field_and_group_draws_agree():
    a, b = prover.StdRng(), R.test_rng()
    n = len(GOLD["discarded"])
    got = synth.from_mont(a.fr(n))
    assert got == [R.fr_rand(b) for _ in range(n)] == [int(x, 0) for x in GOLD["discarded"]]
    assert all(0 <= x < R.R_MOD for x in got)
    beta = synth.from_mont(a.fr(1))[0]
    assert beta == R.fr_rand(b) == int(GOLD["beta"], 0)
    for name in ("g", "gamma_g"):
        p = fq_ints(a.g1())
        assert tuple(p) == R.g1_rand(b) == tuple(int(c, 0) for c in GOLD[name])
        assert (p[1] * p[1] - p[0] ** 3 - 3) % R.Q_MOD == 0
    h = fq_ints(a.g2())
    want = R.g2_rand(b)
    assert h == [want[0][0], want[0][1], want[1][0], want[1][1]] == [int(c, 0) for c in GOLD["h"][0] + GOLD["h"][1]]
    # h lies on the twist and in the subgroup of order r
    assert R.f2_sub(R.f2_mul(want[1], want[1]), R.f2_add(R.f2_mul(R.f2_mul(want[0], want[0]), want[0]), R.G2_B)) == (0, 0)
    assert R.g2_mul(want, R.R_MOD) is None
    # the prover's 27 blinding coefficients follow
    assert synth.from_mont(a.fr(27)) == [R.fr_rand(b) for _ in range(27)] == [int(x, 0) for x in GOLD["blinds"]]


def test_square_roots_in_fq2():
    rng = R.StdRng(bytes(range(32)))
    found = 0
    for _ in range(40):
        x = (R.fq_rand(rng), R.fq_rand(rng))
        sq = R.f2_mul(x, x)
        y = R.f2_sqrt(sq)
        assert y in (x, R.f2_neg(x))
        y2 = R.f2_sqrt(x)
        if y2 is not None:
            found += 1
            assert R.f2_mul(y2, y2) == x
    assert 5 < found < 35
    assert R.f2_sqrt((4, 0)) in ((2, 0), (R.Q_MOD - 2, 0))
    assert R.f2_mul(R.f2_sqrt((R.Q_MOD - 4, 0)), R.f2_sqrt((R.Q_MOD - 4, 0))) == (R.Q_MOD - 4, 0)     # a non-residue of Fq: the root is purely imaginary


def test_opening_challenge_of_test_sponge():
    c = synth.from_mont(prover.test_sponge_challenge().reshape(1, 4))[0]
    assert c == R.opening_challenge() == int(GOLD["opening_challenge"], 0)
    assert c < 1 << 128
    # the permutation is a bijection built from ARK, x^17 and the MDS matrix: inverting the last round by hand recovers its input
    ark = R.poseidon_round_constants()
    assert len(ark) == 39 and all(len(r) == 3 for r in ark)
    out = R.poseidon_permute([0, 0, 0], ark)
    # undo the last MDS: rows (1 0 1), (1 1 0), (0 1 1)  ->  s0 = (o0 + o1 - o2) / 2 ...
    half = pow(2, -1, R.R_MOD)
    s = [(out[0] + out[1] - out[2]) * half % R.R_MOD, (out[1] + out[2] - out[0]) * half % R.R_MOD, (out[0] + out[2] - out[1]) * half % R.R_MOD]
    inv17 = pow(17, -1, R.R_MOD - 1)
    s = [(pow(x, inv17, R.R_MOD) - c38) % R.R_MOD for x, c38 in zip(s, ark[38])]
    assert s == R.poseidon_permute_rounds([0, 0, 0], ark, 38)


def test_replay_of_the_reference_driver_on_the_cpu(orc):
    """src/main.rs:25-230 with the deterministic rng: discarded placeholders, PCS::setup, the proof of the example circuit (CPU oracle
    on the powers of the restated setup, blinding coefficients from the same generator), then verify_plonk_proof_with_transcript on the
    host with the verifier key of that setup (real pairings with h and beta h)."""
    d = R.main_draws()
    s = d["srs"]
    assert [int(c, 0) for c in GOLD["powers_of_g_last"]] == list(s["powers_of_g"][32])
    tables = synth.circuit_tables(*synth.example_circuit())
    g = orc.points_from_ints(s["powers_of_g"]); gg = orc.points_from_ints(s["powers_of_gamma_g"])
    blinds = orc.fr_from_ints(d["blinds"]); chal = orc.fr_from_ints([d["challenge"]])[0]
    from oracle import pymodel as pm
    inp = pm.prepare_inputs(*pm.example_circuit())
    polys = [synth.trim(orc.fr_from_ints(inp[k])) for k in ("a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma")]
    rc, proof, _, _ = orc.prove(g, gg, polys, tables["witness_flat"], tables["sigma"], tables["log_n"], blinds, chal)
    assert rc == 0 and len(proof) == 1170
    assert proof.hex() == GOLD["replay_proof_hex"]
    vk = np.concatenate([orc.points_from_ints([s["g"], s["gamma_g"]]).reshape(-1),
                         orc.fq_from_ints([s["h"][0][0], s["h"][0][1], s["h"][1][0], s["h"][1][1]]).reshape(-1),
                         orc.fq_from_ints([s["beta_h"][0][0], s["beta_h"][0][1], s["beta_h"][1][0], s["beta_h"][1][1]]).reshape(-1)])
    assert prover.verify_vk(vk, proof, chal) is True
    bad = bytearray(proof); bad[3] ^= 4
    assert prover.verify_vk(vk, bytes(bad), chal) is False
    wrong = vk.copy(); wrong[32:48] = vk[16:32]          # beta_h := h
    assert prover.verify_vk(wrong, proof, chal) is False
