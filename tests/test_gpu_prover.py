"""Whole proofs: MI355X host prover + HIP library vs the CPU oracle, byte for byte, on identical inputs."""
import numpy as np
import pytest

from conftest import assert_polys_equal, independent_polys
from oracle import pymodel as pm
from plonk_amd import synth

pytestmark = pytest.mark.gpu

TAU, GSCAL = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF012345, 0x0FEDCBA9876543210FEDCBA987654321


@pytest.fixture(scope="module")
def pk(orc):
    from plonk_amd.prover import ProverKey
    k = ProverKey(0)
    yield k
    k.close()


def _prove_both(pk, orc, variables, gates, seed):
    tables = synth.circuit_tables(variables, gates)
    n, log_n = tables["n"], tables["log_n"]
    n4 = 1 << (3 * n - 1).bit_length()
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, n4 + 1)
    rng = pm.SplitMix64(seed)
    blinds = orc.fr_from_ints([rng.next_fr() for _ in range(27)])
    chal = orc.fr_from_ints([rng.next_u64() | (rng.next_u64() << 64)])[0]
    pk.set_opening_challenge(chal)
    inputs = pk.prepare(tables)
    res = pk.prove(inputs, blinds)
    # the checker's inputs come from the CPU alone: interpolated polynomials and SRS powers of the oracle, compared with the device's
    polys = independent_polys(orc, tables)
    assert_polys_equal(inputs.polys, polys)
    one = orc.fr_from_ints([1])[0]
    g = orc.srs_powers(tau, one, 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    assert np.array_equal(pk.ctx.srs_read(0, 0, n4 + 1), g) and np.array_equal(pk.ctx.srs_read(1, 0, 3), gg)
    rc, want, ch, _ = orc.prove(g, gg, polys, tables["witness_flat"], tables["sigma"], log_n, blinds, chal)
    assert rc == 0
    inputs.close()
    return res, want, ch


def test_example_circuit_proof_bytes(pk, orc):
    """src/main.rs:29-39, n = 4, N4 = 16."""
    res, want, ch = _prove_both(pk, orc, *synth.example_circuit(), seed=3)
    assert np.array_equal(res.challenges, ch)
    assert res.proof_bytes == want
    assert len(res.proof_bytes) == 1170


@pytest.mark.parametrize("log_n", [1, 3, 6, 10, 12])
def test_synthetic_proof_bytes(pk, orc, log_n):
    res, want, ch = _prove_both(pk, orc, *synth.synthetic_circuit(log_n, seed=log_n), seed=50 + log_n)
    assert np.array_equal(res.challenges, ch)
    assert res.proof_bytes == want


def test_proof_is_deterministic_and_inputs_reusable(pk, orc):
    variables, gates = synth.synthetic_circuit(8, seed=9)
    tables = synth.circuit_tables(variables, gates)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    pk.set_opening_challenge(orc.fr_from_ints([12345678901234567890])[0])
    blinds = synth.seeded_scalars(27, 4)
    inputs = pk.prepare(tables)
    p1 = pk.prove(inputs, blinds).proof_bytes
    p2 = pk.prove(inputs, blinds).proof_bytes
    p3 = pk.prove(inputs, synth.seeded_scalars(27, 5)).proof_bytes
    assert p1 == p2 and p1 != p3
    inputs.close()


def test_broken_witness_is_rejected(pk, orc):
    """assert!(remainder.is_zero(), "t(X) not divisible by Z_H(X)") src/circuit.rs:487."""
    from plonk_amd.hip import PlonkHipError
    variables, gates = synth.synthetic_circuit(5, seed=2)
    tables = synth.circuit_tables(variables, gates)
    tables["c"] = tables["c"].copy(); tables["c"][3] = tables["a"][0]
    tables["witness_flat"] = np.stack([tables["a"], tables["b"], tables["c"]], axis=1).reshape(-1, 4)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    inputs = pk.prepare(tables)
    with pytest.raises(PlonkHipError) as e:
        pk.prove(inputs, synth.seeded_scalars(27, 4))
    assert e.value.status == "PLK_ERR_NOT_DIVISIBLE" and "not divisible by Z_H" in str(e.value)
    inputs.close()


@pytest.mark.parametrize("log_n", [2, 14])
def test_gpu_proof_passes_the_pairing_verifier(pk, orc, log_n):
    """src/main.rs:211-230: prove, then verify_plonk_proof_with_transcript.  The verifier runs on the CPU with real pairings, so
    this needs neither the oracle nor the SRS trapdoor identities the other tests use."""
    from plonk_amd import prover
    variables, gates = synth.example_circuit() if log_n == 2 else synth.synthetic_circuit(log_n, seed=77)
    tables = synth.circuit_tables(variables, gates)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    chal = orc.fr_from_ints([0x1234567890ABCDEF1122334455667788])[0]
    pk.set_opening_challenge(chal)
    inputs = pk.prepare(tables)
    proof = pk.prove(inputs, synth.seeded_scalars(27, 6)).proof_bytes
    inputs.close()
    assert prover.verify(tau, gs, proof, chal) is True
    bad = bytearray(proof); bad[40] ^= 0x10
    assert prover.verify(tau, gs, bytes(bad), chal) is False


def _setup_points(orc, s):
    g = orc.points_from_ints(s["powers_of_g"]); gg = orc.points_from_ints(s["powers_of_gamma_g"])
    return g, gg


def test_setup_powers_on_the_device(pk, orc):
    """plk_srs_setup: PCS::setup + trim (src/main.rs:182-183) from the rng draws beta, g, gamma_g, against the CPU restatement that
    multiplies point by point.  g and gamma_g are arbitrary curve points here, not multiples of (1, 2) with a known factor."""
    from oracle import refrng as R
    s = R.main_draws()["srs"]
    want_g, want_gg = _setup_points(orc, s)
    beta = orc.fr_from_ints([s["beta"]])[0]
    g0, gamma0 = orc.points_from_ints([s["g"]])[0], orc.points_from_ints([s["gamma_g"]])[0]
    pk.ctx.srs_setup(beta, g0, gamma0, 33)
    assert np.array_equal(pk.ctx.srs_read(0, 0, 33), want_g)
    assert np.array_equal(pk.ctx.srs_read(1, 0, 3), want_gg)
    # block-cyclic shards of the same setup: every rank reads back its own blocks
    for rank in range(2):
        pk.ctx.srs_setup(beta, g0, gamma0, 33, rank=rank, world=2, log_block=2)
        for blk in range(rank, 9, 2):
            first, cnt = 4 * blk, min(4, 33 - 4 * blk)
            assert np.array_equal(pk.ctx.srs_read(0, first, cnt), want_g[first:first + cnt])
    # a point off the curve is refused
    from plonk_amd.hip import PlonkHipError
    bad = g0.copy(); bad[0] ^= 1
    with pytest.raises(PlonkHipError):
        pk.ctx.srs_setup(beta, bad, gamma0, 33)


def test_setup_from_the_generator_equals_generate(pk, orc):
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 5000)
    want_g, want_gg = pk.ctx.srs_read(0, 0, 5000), pk.ctx.srs_read(1, 0, 3)
    gen = orc.points_from_ints([pm.G1_GEN])[0]
    pk.ctx.srs_setup(tau, gen, want_gg[0], 5000)
    assert np.array_equal(pk.ctx.srs_read(0, 0, 5000), want_g)
    assert np.array_equal(pk.ctx.srs_read(1, 0, 3), want_gg)


def test_replay_of_the_reference_driver(pk, orc):
    """src/main.rs:25-230 end to end with the reference's own randomness (ark_std::test_rng, deterministic branch): the placeholder
    draws, PCS::setup on the GPU, the proof of the example circuit with its 27 blinding coefficients drawn from the same generator and
    the opening challenge of test_sponge, then the pairing verifier with the key of that setup.  Bytes are compared with the CPU oracle
    run on the Python restatement of the same draws."""
    from oracle import refrng as R
    from plonk_amd import prover
    tables = synth.circuit_tables(*synth.example_circuit())
    rng = prover.StdRng()
    rng.fr(2 * 3 * tables["n"] + 3)                       # src/main.rs:136-137,174-176
    vk = pk.setup(32, rng)                                # src/main.rs:182-183
    chal = prover.test_sponge_challenge()
    pk.set_opening_challenge(chal)
    inputs = pk.prepare(tables)
    res = pk.prove_rng(inputs, rng)                       # src/main.rs:211
    assert prover.verify_vk(vk, res.proof_bytes, chal) is True          # src/main.rs:226
    d = R.main_draws()
    s = d["srs"]
    g, gg = _setup_points(orc, s)
    assert np.array_equal(vk[:16].reshape(2, 8), orc.points_from_ints([s["g"], s["gamma_g"]]))
    rc, want, ch, _ = orc.prove(g, gg, inputs.polys, tables["witness_flat"], tables["sigma"], tables["log_n"], orc.fr_from_ints(d["blinds"]),
                                orc.fr_from_ints([d["challenge"]])[0])
    inputs.close()
    assert rc == 0 and res.proof_bytes == want and np.array_equal(res.challenges, ch)
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refrng.json")))
    assert res.proof_bytes.hex() == gold["replay_proof_hex"]
    # the generator has moved on by exactly the prover's 27 draws
    after = R.test_rng()
    for _ in range(2 * 3 * tables["n"] + 3): R.fr_rand(after)
    R.kzg_setup(32, after)
    for _ in range(27): R.fr_rand(after)
    assert rng.next_u64() == after.next_u64()


def test_proof_with_public_inputs(pk, orc):
    """absorb_public_inputs (src/prover.rs:138, src/transcript.rs:48-52): the host prover and the verifier absorb the same message as the oracle."""
    from plonk_amd import prover
    variables, gates = synth.synthetic_circuit(5, seed=21)
    tables = synth.circuit_tables(variables, gates)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    chal = orc.fr_from_ints([0xFEEDFACECAFEBEEF0123456789ABCDEF])[0]
    pk.set_opening_challenge(chal)
    blinds = synth.seeded_scalars(27, 8)
    pub = orc.fr_from_ints([57, pm.R_MOD - 1, 0])
    inputs = pk.prepare(tables)
    res = pk.prove(inputs, blinds, public_inputs=pub)
    plain = pk.prove(inputs, blinds)
    # the checker's inputs are built on the CPU alone and compared with the device's before they are used
    polys = independent_polys(orc, tables)
    assert_polys_equal(inputs.polys, polys)
    n4 = 4 * tables["n"]
    g = orc.srs_powers(tau, orc.fr_from_ints([1])[0], 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    assert np.array_equal(pk.ctx.srs_read(0, 0, n4 + 1), g) and np.array_equal(pk.ctx.srs_read(1, 0, 3), gg)
    rc, want, ch, _ = orc.prove(g, gg, polys, tables["witness_flat"], tables["sigma"], tables["log_n"], blinds, chal, public_inputs=pub)
    inputs.close()
    assert rc == 0 and res.proof_bytes == want and np.array_equal(res.challenges, ch)
    assert plain.proof_bytes != res.proof_bytes
    assert prover.verify(tau, gs, res.proof_bytes, chal, public_inputs=pub) is True
    assert prover.verify(tau, gs, res.proof_bytes, chal) is False                      # verifier replays a different transcript
    assert prover.verify(tau, gs, res.proof_bytes, chal, public_inputs=pub[:2]) is False


def test_prover_failure_behaviour(pk, orc):
    """The reference's panics as plonk::Error / PlonkHipError: a wire polynomial with fewer than n coefficients (trailing zero values are
    trimmed by from_coefficients_vec, then a_col[i] indexes out of bounds, src/circuit.rs:446), an SRS shorter than a polynomial
    (TooManyCoefficients, src/prover.rs:150), an rng that runs dry."""
    from plonk_amd.hip import PlonkHipError
    variables, gates = synth.synthetic_circuit(4, seed=31)
    tables = synth.circuit_tables(variables, gates)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    pk.set_opening_challenge(orc.fr_from_ints([77])[0])
    good = pk.prepare(tables)
    blinds = synth.seeded_scalars(27, 4)
    ok = pk.prove(good, blinds).proof_bytes
    # 26 blinding scalars: the ninth commitment cannot draw its third coefficient
    with pytest.raises(PlonkHipError) as e:
        pk.prove(good, blinds[:26])
    assert "rng exhausted" in str(e.value)
    # the key still proves afterwards, with the same bytes
    assert pk.prove(good, blinds).proof_bytes == ok
    # a wire column whose last value is zero loses a coefficient
    short = [p.copy() for p in good.polys]
    short[1] = short[1][:-1]
    inputs = pk.upload(short, tables["witness_flat"], tables["sigma"], tables["log_n"])
    with pytest.raises(PlonkHipError) as e:
        pk.prove(inputs, blinds)
    assert e.value.status == "PLK_ERR_INDEX_OUT_OF_RANGE" and "index out of bounds" in str(e.value)
    inputs.close()
    # a key with fewer powers than s_sigma has coefficients
    pk.generate_srs(tau, gs, 2 * tables["n"])
    inputs = pk.upload(good.polys, tables["witness_flat"], tables["sigma"], tables["log_n"])
    with pytest.raises(PlonkHipError) as e:
        pk.prove(inputs, blinds)
    assert e.value.status == "PLK_ERR_SRS_TOO_SHORT" and "TooManyCoefficients" in str(e.value)
    inputs.close()
    good.close()
    # and it recovers once the key is long enough again
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    again = pk.prepare(tables)
    assert pk.prove(again, blinds).proof_bytes == ok
    again.close()


def test_cxx_replay_of_the_reference_driver(pk, orc):
    """examples/replay_main.cpp: src/main.rs in C++ on the host library (CircuitBuilder, setup from test_rng, proof, pairing
    verification).  Its proof must be the one the Python replay above produces."""
    import os
    import subprocess
    from plonk_amd import prover
    exe = os.path.join(os.path.dirname(prover.HOST_LIB_PATH), "replay_main")
    assert os.path.exists(exe), "build it with make -C plonk_amd/host"
    out = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    assert "Proof verified: true" in lines and "Circuit gates: 4" in lines and "Domain size: 4" in lines
    tables = synth.circuit_tables(*synth.example_circuit())
    rng = prover.StdRng()
    rng.fr(2 * 3 * tables["n"] + 3)
    pk.setup(32, rng)
    pk.set_opening_challenge(prover.test_sponge_challenge())
    inputs = pk.prepare(tables)
    want = pk.prove_rng(inputs, rng).proof_bytes
    inputs.close()
    assert lines[0] == "proof " + want.hex()


@pytest.mark.parametrize("log_n", [3, 10, 14, 19])
def test_proof_from_host_arrays_equals_the_resident_one(pk, orc, log_n):
    """create_plonk_proof_with_transcript on borrowed host arrays (src/prover.rs:119-133): the polynomials are uploaded by the prover
    itself while its first commitments run.  Same bytes as the proof of inputs that were resident beforehand, from ordinary and from
    page-locked arrays, and again after a proof of another size has used the same device buffers.  At 2^19 gates the ordinary arrays
    go several times round the library's ring of staging buffers (64 MB of s_sigma, 48 MB of witness values)."""
    variables, gates = synth.synthetic_circuit(log_n, seed=40 + log_n)
    tables = synth.circuit_tables(variables, gates)
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    pk.generate_srs(tau, gs, 4 * tables["n"] + 1)
    pk.set_opening_challenge(synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
    blinds = synth.seeded_scalars(27, 9)
    inputs = pk.prepare(tables)
    want = pk.prove(inputs, blinds)
    # host arrays made by the CPU oracle alone (not read back from the device), and the oracle's own proof of them
    own = independent_polys(orc, tables)
    assert_polys_equal(inputs.polys, own)
    polys = [np.ascontiguousarray(p) for p in own]
    wflat = np.ascontiguousarray(tables["witness_flat"], dtype=np.uint64); sigma = np.ascontiguousarray(tables["sigma"], dtype=np.uint64)
    got = pk.prove_host(polys, wflat, sigma, tables["log_n"], blinds)
    assert got.proof_bytes == want.proof_bytes and np.array_equal(got.challenges, want.challenges)
    if log_n <= 14:   # the oracle's MSMs take minutes beyond that; 2^19 gates stay pinned through the resident proof (trapdoor tests cover that size)
        n4 = 4 * tables["n"]
        g = orc.srs_powers(tau, orc.fr_from_ints([1])[0], 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
        assert np.array_equal(pk.ctx.srs_read(0, 0, n4 + 1), g) and np.array_equal(pk.ctx.srs_read(1, 0, 3), gg)
        rc, oracle_bytes, ch, _ = orc.prove(g, gg, own, tables["witness_flat"], tables["sigma"], tables["log_n"], blinds,
                                            synth.to_mont([0x1234567890ABCDEF1122334455667788])[0])
        assert rc == 0 and got.proof_bytes == oracle_bytes and np.array_equal(got.challenges, ch)
    # trailing zeros are stripped as DensePolynomial::from_coefficients_vec does
    padded = [np.concatenate([p, np.zeros((3, 4), np.uint64)]) for p in polys]
    assert pk.prove_host(padded, wflat, sigma, tables["log_n"], blinds).proof_bytes == want.proof_bytes
    pinned = []
    for a in polys + [wflat, sigma]:
        h = pk.ctx.host_empty(a.shape, a.dtype); h[...] = a; pinned.append(h)
    got = pk.prove_host(pinned[:7], pinned[7], pinned[8], tables["log_n"], blinds)
    assert got.proof_bytes == want.proof_bytes
    for h in pinned:
        pk.ctx.host_free(h)
    inputs.close()


def test_two_keys_driven_by_two_threads(orc):
    """Two prover keys on one GPU, each driven by its own host thread (the keys share the device's streams, everything else is per
    key): every proof of either thread has the bytes the oracle produces."""
    import threading
    from plonk_amd.prover import ProverKey
    log_n = 12
    tables = synth.circuit_tables(*synth.synthetic_circuit(log_n, seed=9))
    n4 = 4 * tables["n"]
    tau = orc.fr_from_ints([TAU])[0]; gs = orc.fr_from_ints([GSCAL])[0]
    blinds = synth.seeded_scalars(27, 21)
    chal = synth.to_mont([0x5EED5EED5EED5EED1234])[0]
    keys = []
    for _ in range(2):
        k = ProverKey(0)
        k.generate_srs(tau, gs, n4 + 1)
        k.set_opening_challenge(chal)
        keys.append((k, k.prepare(tables)))
    polys = independent_polys(orc, tables)
    one = orc.fr_from_ints([1])[0]
    g = orc.srs_powers(tau, one, 0, n4 + 1); gg = orc.srs_powers(tau, gs, 0, 3)
    rc, want, _, _ = orc.prove(g, gg, polys, tables["witness_flat"], tables["sigma"], log_n, blinds, chal)
    assert rc == 0
    results = [[], []]

    def run(i):
        k, inputs = keys[i]
        for _ in range(12):
            results[i].append(k.prove(inputs, blinds).proof_bytes)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=100)
    assert not any(t.is_alive() for t in threads), "a prover thread did not finish"
    for i in range(2):
        assert len(results[i]) == 12 and all(p == want for p in results[i])
    for k, inputs in keys:
        inputs.close(); k.close()
