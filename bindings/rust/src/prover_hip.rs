//! Drop-in bodies for the reference's hot path on top of `hip.rs`.  Add `mod hip; mod prover_hip;` next to the other `mod` lines of
//! src/main.rs:1-5 and call `prover_hip::create_plonk_proof_with_transcript` at src/main.rs:211 (same arguments plus the backend).
//!
//! Signatures, rng draw order, transcript absorb order and panics follow src/prover.rs:119-336, src/circuit.rs:491-544 and
//! src/poly_utils.rs:7-31.  What changes is where the arithmetic runs: `PCS::commit` / `PCS::open` / `build_grand_product` /
//! `build_quotient_polynomial` / `fft` become calls into libplonk_hip; the merlin transcript, the Poseidon sponge, the rng and
//! serialization stay here on the CPU.  The dead `s_sigma_vals` evaluation (src/prover.rs:240, never read) and the debug prints
//! are not reproduced.  NOT COMPILED in this repository (no Rust toolchain); tested equivalent: plonk_amd/host/prover.cpp.
use ark_bn254::{Bn254, Fr, G1Affine};
use ark_crypto_primitives::sponge::{CryptographicSponge, FieldElementSize};
use ark_ff::UniformRand;
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, GeneralEvaluationDomain};
use ark_poly_commit::{kzg10, marlin::marlin_pc};
use ark_serialize::CanonicalSerialize;
use ark_std::rand::Rng;

use crate::hip::Backend;
use crate::prover::{KZGCommitment, KZGOpeningProof, KZGProverKey, PlonkProof};
use crate::transcript::{absorb_public_inputs, PlonkTranscript};

/// `PCS::commit(pk, [&labeled], Some(rng))` for one polynomial with hiding bound 1 and no degree bound.
/// kzg10 draws the blinding polynomial (degree hiding_bound + 1 = 2, three `Fr::rand`) after the main MSM; only the order of
/// draws across commitments is observable, and it is kept.
fn commit<R: Rng>(hip: &Backend, poly: &DensePolynomial<Fr>, rng: &mut R) -> (KZGCommitment, [Fr; 3]) {
    let blind = [Fr::rand(rng), Fr::rand(rng), Fr::rand(rng)];
    let point: G1Affine = hip.commit(&poly.coeffs, &blind);
    (marlin_pc::Commitment { comm: kzg10::Commitment(point), shifted_comm: None }, blind)
}

/// `PCS::open(pk, [&labeled], &comms, &zeta, &mut sponge, &states, None)` for one polynomial: marlin_pc multiplies polynomial and
/// randomness by one 128-bit challenge from the sponge even when there is a single polynomial.
fn open<S: CryptographicSponge>(hip: &Backend, poly: &DensePolynomial<Fr>, blind: &[Fr; 3], zeta: &Fr, sponge: &mut S) -> (KZGOpeningProof, Fr) {
    let challenge: Fr = sponge.squeeze_field_elements_with_sizes(&[FieldElementSize::Truncated(128)])[0];
    let (w, random_v, eval) = hip.open(&poly.coeffs, blind, zeta, &challenge);
    (kzg10::Proof::<Bn254> { w, random_v: Some(random_v) }, eval)
}

fn absorb_commitment(transcript: &mut PlonkTranscript, label: &'static [u8], c: &KZGCommitment) {
    let mut buf = Vec::new();
    c.serialize_compressed(&mut buf).unwrap();
    transcript.inner().append_message(label, &buf);
}

/// `Circuit::<Fr>::build_grand_product` (src/circuit.rs:491-544).  `s_id_vals` must be the labels the prover passes (src/prover.rs:205).
pub fn build_grand_product(hip: &Backend, witness_flat: &[Fr], sigma: &[usize], domain: GeneralEvaluationDomain<Fr>, beta: Fr, gamma: Fr,
                           s_id_vals: &[Fr]) -> DensePolynomial<Fr> {
    debug_assert!(s_id_vals.iter().enumerate().all(|(i, v)| *v == Fr::from(i as u64)));
    DensePolynomial::from_coefficients_vec(hip.grand_product(witness_flat, sigma, domain.size(), &beta, &gamma))
}

/// `interpolate_selector` (src/poly_utils.rs:7-11).
pub fn interpolate_selector(hip: &Backend, values: &[Fr]) -> DensePolynomial<Fr> {
    let mut v = values.to_vec();
    v.resize(values.len().next_power_of_two(), Fr::from(0u64));
    hip.ntt(&mut v, true);
    DensePolynomial::from_coefficients_vec(v)
}

#[allow(clippy::too_many_arguments)]
pub fn create_plonk_proof_with_transcript<R: Rng>(
    hip: &Backend, // the SRS of `pk` has been loaded with Backend::load_srs(&pk.powers, &pk.powers_of_gamma_g)
    _pk: &KZGProverKey,
    a: &DensePolynomial<Fr>,
    b: &DensePolynomial<Fr>,
    c: &DensePolynomial<Fr>,
    q_add: &DensePolynomial<Fr>,
    q_mul: &DensePolynomial<Fr>,
    s_id: &DensePolynomial<Fr>,
    s_sigma: &DensePolynomial<Fr>,
    public_inputs: &[Fr],
    witness_flat: &[Fr],
    sigma: &[usize],
    domain: &GeneralEvaluationDomain<Fr>,
    rng: &mut R,
) -> PlonkProof<Fr> {
    let n = domain.size();
    let mut transcript = PlonkTranscript::new(b"plonk_proof");
    absorb_public_inputs(transcript.inner(), public_inputs);

    // round 1: commitments in the order a, b, c, q_add, q_mul, s_id, s_sigma (the rng order), absorbed as q_add, q_mul, a, b, c, s_id, s_sigma
    let (comm_a, r_a) = commit(hip, a, rng);
    let (comm_b, r_b) = commit(hip, b, rng);
    let (comm_c, r_c) = commit(hip, c, rng);
    let (comm_q_add, r_q_add) = commit(hip, q_add, rng);
    let (comm_q_mul, r_q_mul) = commit(hip, q_mul, rng);
    let (comm_s_id, r_s_id) = commit(hip, s_id, rng);
    let (comm_s_sigma, r_s_sigma) = commit(hip, s_sigma, rng);
    absorb_commitment(&mut transcript, b"q_add", &comm_q_add);
    absorb_commitment(&mut transcript, b"q_mul", &comm_q_mul);
    absorb_commitment(&mut transcript, b"a", &comm_a);
    absorb_commitment(&mut transcript, b"b", &comm_b);
    absorb_commitment(&mut transcript, b"c", &comm_c);
    absorb_commitment(&mut transcript, b"s_id", &comm_s_id);
    absorb_commitment(&mut transcript, b"s_sigma", &comm_s_sigma);

    // round 2
    let beta = transcript.challenge_beta::<Fr>();
    let gamma = transcript.challenge_gamma::<Fr>();
    let s_id_vals: Vec<Fr> = (0..3 * n).map(|i| Fr::from(i as u64)).collect();
    let z_poly = build_grand_product(hip, witness_flat, sigma, domain.clone(), beta, gamma, &s_id_vals);
    let (comm_z, r_z) = commit(hip, &z_poly, rng);
    absorb_commitment(&mut transcript, b"z", &comm_z);

    // round 3: alpha is squeezed (it advances the transcript) but does not enter the arithmetic: the temporary circuit of
    // src/prover.rs:224-227 has no public inputs, which is the only place alpha is used (src/circuit.rs:432)
    let _alpha = transcript.challenge_alpha::<Fr>();
    let mut q_add_vals = q_add.coeffs.clone();
    let mut q_mul_vals = q_mul.coeffs.clone();
    let mut z_vals = z_poly.coeffs.clone();
    for v in [&mut q_add_vals, &mut q_mul_vals, &mut z_vals] {
        v.resize(n, Fr::from(0u64)); // domain.fft zero-pads
        hip.ntt(v, false);
    }
    // a, b, c are used as columns of VALUES here although they are coefficient vectors (src/prover.rs:229-231)
    let t_coeffs = hip.quotient(&a.coeffs, &b.coeffs, &c.coeffs, &q_add_vals, &q_mul_vals, &z_vals, sigma, n, &beta, &gamma);
    let t_poly = DensePolynomial::from_coefficients_vec(t_coeffs);
    let (comm_t, r_t) = commit(hip, &t_poly, rng);
    absorb_commitment(&mut transcript, b"t", &comm_t);

    // rounds 4 and 5: the opening's division by (X - zeta) leaves f(zeta) as its remainder, so every evaluation comes back with
    // its opening; the transcript still absorbs the nine evaluations in the reference's order (nothing is squeezed afterwards)
    let zeta = transcript.challenge_zeta::<Fr>();
    let mut sponges: Vec<_> = (0..9).map(|_| crate::prover::test_sponge::<Fr>()).collect(); // make test_sponge pub(crate)
    let polys = [a, b, c, q_add, q_mul, s_id, s_sigma, &z_poly, &t_poly];
    let blinds = [r_a, r_b, r_c, r_q_add, r_q_mul, r_s_id, r_s_sigma, r_z, r_t];
    let mut proofs = Vec::with_capacity(9);
    let mut evals = Vec::with_capacity(9);
    for i in 0..9 {
        let (p, e) = open(hip, polys[i], &blinds[i], &zeta, &mut sponges[i]);
        proofs.push(p);
        evals.push(e);
    }
    transcript.absorb_evaluations(&evals[0], &evals[1], &evals[2], &evals[3], &evals[4], &evals[5], &evals[6], &evals[7], &evals[8]);

    let mut proofs = proofs.into_iter();
    let mut next = || proofs.next().unwrap();
    PlonkProof {
        eval_a: evals[0], eval_b: evals[1], eval_c: evals[2], eval_q_add: evals[3], eval_q_mul: evals[4],
        eval_s_id: evals[5], eval_s_sigma: evals[6], eval_z: evals[7], eval_t: evals[8],
        comm_a, comm_b, comm_c, comm_q_add, comm_q_mul, comm_s_id, comm_s_sigma, comm_z, comm_t,
        proof_a: next(), proof_b: next(), proof_c: next(), proof_q_add: next(), proof_q_mul: next(),
        proof_s_id: next(), proof_s_sigma: next(), proof_z: next(), proof_t: next(),
    }
}
