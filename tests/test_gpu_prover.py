"""Whole proofs: MI355X host prover + HIP library vs the CPU oracle, byte for byte, on identical inputs."""
import numpy as np
import pytest

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
    g = pk.ctx.srs_read(0, 0, n4 + 1); gg = pk.ctx.srs_read(1, 0, 3)
    rc, want, ch, _ = orc.prove(g, gg, inputs.polys, tables["witness_flat"], tables["sigma"], log_n, blinds, chal)
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
