//! `create_plonk_proof_with_transcript` with the reference's exact signature (src/prover.rs:119-133) on the whole-proof entry point of
//! libplonk_host (include/plonk_host.h: plonk_prove_host).  Replace the body in src/prover.rs by a call to this function, or add
//! `mod prover_fast;` next to the `mod` lines of src/main.rs:1-5 and change the path at the call site src/main.rs:211; the arguments
//! stay as they are:
//!
//! ```ignore
//! let proof = prover_fast::create_plonk_proof_with_transcript(
//!     &pk, &a_poly, &b_poly, &c_poly, &q_add_poly, &q_mul_poly, &s_id_poly, &s_sigma_poly,
//!     &public_inputs, &witness_flat, &sigma, &domain, &mut rng);          // src/main.rs:211-221, unchanged
//! ```
//!
//! What stays on the CPU, exactly as in the reference: the rng (the 27 `Fr::rand` draws of the nine hiding commitments, in the order
//! kzg10::commit makes them) and the Poseidon opening challenge of `test_sponge` (src/prover.rs:36-59).  Everything else - the merlin
//! transcript included, which libplonk_host restates - runs behind one FFI call with the polynomials resident in HBM, the
//! commitments / openings of a round in flight together, and the host vectors crossing PCIe while the first commitments are
//! computed.  The proof comes back as the ark-serialize compressed bytes of the 27 fields of `PlonkProof` and is deserialized here.
//!
//! The GPU side of a `KZGProverKey` (its powers in HBM plus window tables) is created on first use of that key and kept for the
//! life of the process, keyed by the address of the key's `powers` vector.
//! NOT COMPILED in this repository (no Rust toolchain); the C++ host prover it calls is what the tests drive.
use std::cell::RefCell;
use std::ffi::{c_void, CStr};

use ark_bn254::{Bn254, Fr, G1Affine};
use ark_crypto_primitives::sponge::{CryptographicSponge, FieldElementSize};
use ark_ff::UniformRand;
use ark_poly::{univariate::DensePolynomial, EvaluationDomain, GeneralEvaluationDomain};
use ark_poly_commit::{kzg10, marlin::marlin_pc};
use ark_serialize::CanonicalDeserialize;
use ark_std::rand::Rng;

use crate::prover::{KZGCommitment, KZGOpeningProof, KZGProverKey, PlonkProof};

// include/plonk_host.h
extern "C" {
    fn plonk_host_last_error() -> *const std::os::raw::c_char;
    fn plonk_pk_create(device: i32, pk_out: *mut *mut c_void) -> i32;
    fn plonk_pk_destroy(pk: *mut c_void) -> i32;
    fn plonk_pk_load(pk: *mut c_void, g_xy: *const u64, len: u64, gamma_g_xy: *const u64) -> i32;
    fn plonk_pk_set_opening_challenge(pk: *mut c_void, challenge: *const u64) -> i32;
    fn plonk_prove_host(pk: *mut c_void, polys: *const *const u64, lens: *const u64, witness_flat: *const u64, sigma: *const u64, log_n: u32,
                        public_inputs: *const u64, n_public: u64, blinds: *const u64, n_blinds: u64, proof_bytes: *mut u8, challenges: *mut u64,
                        seconds: *mut f64) -> i32;
}

/// The reference has no `Result`s on this path: every failure is a panic, with the reference's own message where it has one.
fn check(rc: i32) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(plonk_host_last_error()) }.to_string_lossy().into_owned();
        panic!("{}", msg);
    }
}

/// Identity of a key by CONTENT, not by address: an address is reused once a key has been dropped, and a stale device SRS behind a
/// new key would give wrong proofs without any error.  Length plus the first, the middle and the last power (three distinct
/// multiples of one generator: two different SRS do not share them) plus the first gamma power.
#[derive(Clone, PartialEq, Eq, Hash)]
struct KeyId {
    len: usize,
    probe: [[u64; 8]; 4],
}
fn key_id(pk: &KZGProverKey) -> KeyId {
    let xy = |p: &G1Affine| -> [u64; 8] {
        let mut o = [0u64; 8];
        o[..4].copy_from_slice(&(p.x.0).0);
        o[4..].copy_from_slice(&(p.y.0).0);
        o
    };
    let n = pk.powers.len();
    assert!(n > 0 && !pk.powers_of_gamma_g.is_empty());
    KeyId { len: n, probe: [xy(&pk.powers[0]), xy(&pk.powers[n / 2]), xy(&pk.powers[n - 1]), xy(&pk.powers_of_gamma_g[0])] }
}
/// Device side of one key; released (SRS tables, scratch, streams) when the owning thread ends or the cache evicts it.
struct GpuKey(*mut c_void);
impl Drop for GpuKey {
    fn drop(&mut self) {
        unsafe { plonk_pk_destroy(self.0) };
    }
}
const MAX_CACHED_KEYS: usize = 4; // a 2^22-point SRS holds about 10 GB of HBM with its tables and scratch

thread_local! {
    // the reference is single threaded; a key's GPU side belongs to the thread that first used it
    static KEYS: RefCell<Vec<(KeyId, GpuKey)>> = RefCell::new(Vec::new());
}

/// GPU side of `pk`, created on first use: `powers` and `powers_of_gamma_g` of marlin_pc::CommitterKey go to HBM (PCS::trim's
/// output, src/main.rs:183), the opening challenge of `test_sponge` is computed once.  Least recently used keys are dropped
/// beyond MAX_CACHED_KEYS.
fn gpu_key(pk: &KZGProverKey) -> *mut c_void {
    let id = key_id(pk);
    KEYS.with(|keys| {
        let mut keys = keys.borrow_mut();
        if let Some(pos) = keys.iter().position(|(k, _)| *k == id) {
            let entry = keys.remove(pos);
            let h = (entry.1).0;
            keys.push(entry); // most recently used last
            return h;
        }
        if keys.len() >= MAX_CACHED_KEYS {
            keys.remove(0); // Drop releases its device memory
        }
        let h = {
            let mut h: *mut c_void = std::ptr::null_mut();
            check(unsafe { plonk_pk_create(0, &mut h) });
            assert!(pk.powers_of_gamma_g.len() >= 3, "hiding bound 1 needs three powers of gamma_g");
            let limbs = |pts: &[G1Affine]| -> Vec<u64> {
                let mut out = Vec::with_capacity(8 * pts.len());
                for p in pts {
                    assert!(!p.infinity, "SRS powers are never the identity");
                    out.extend_from_slice(&(p.x.0).0);
                    out.extend_from_slice(&(p.y.0).0);
                }
                out
            };
            let (g, gg) = (limbs(&pk.powers), limbs(&pk.powers_of_gamma_g[..3]));
            check(unsafe { plonk_pk_load(h, g.as_ptr(), pk.powers.len() as u64, gg.as_ptr()) });
            // every PCS::open of the reference squeezes this one value from a fresh test_sponge() (src/prover.rs:285-303)
            let mut sponge = crate::prover::test_sponge::<Fr>(); // make test_sponge pub(crate)
            let c: Fr = sponge.squeeze_field_elements_with_sizes(&[FieldElementSize::Truncated(128)])[0];
            check(unsafe { plonk_pk_set_opening_challenge(h, (c.0).0.as_ptr()) });
            h
        };
        keys.push((id, GpuKey(h)));
        h
    })
}

#[allow(clippy::too_many_arguments)]
pub fn create_plonk_proof_with_transcript<R: Rng>(
    pk: &KZGProverKey,
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
    assert!(witness_flat.len() >= 3 * n && sigma.len() >= 3 * n, "index out of bounds: witness_flat / sigma shorter than 3n");
    let h = gpu_key(pk);
    // kzg10::commit draws its blinding polynomial (degree hiding_bound + 1 = 2: three coefficients) per commitment, in the order of
    // the commitments: a, b, c, q_add, q_mul, s_id, s_sigma (src/prover.rs:150-156), z (:210), t (:253).  Nothing else on this path
    // touches the rng, so drawing all 27 up front leaves `rng` in the state the reference leaves it in.
    let blinds: Vec<Fr> = (0..27).map(|_| Fr::rand(rng)).collect();
    // Fp256<MontBackend<_, 4>> is #[repr(transparent)]-like over BigInt<4> = [u64; 4] in Montgomery form: slices of Fr are passed as they are
    let polys: [*const u64; 7] = [a, b, c, q_add, q_mul, s_id, s_sigma].map(|p| p.coeffs.as_ptr() as *const u64);
    let lens: [u64; 7] = [a, b, c, q_add, q_mul, s_id, s_sigma].map(|p| p.coeffs.len() as u64);
    const _: () = assert!(std::mem::size_of::<usize>() == 8); // sigma: &[usize] is read as u64 indices
    let mut bytes = [0u8; 1170];
    check(unsafe {
        plonk_prove_host(h, polys.as_ptr(), lens.as_ptr(), witness_flat.as_ptr() as *const u64, sigma.as_ptr() as *const u64, n.trailing_zeros(),
                         public_inputs.as_ptr() as *const u64, public_inputs.len() as u64, blinds.as_ptr() as *const u64, 27, bytes.as_mut_ptr(),
                         std::ptr::null_mut(), std::ptr::null_mut())
    });
    // 9 evaluations, 9 commitments, 9 opening proofs in the field order of PlonkProof (src/prover.rs:61-95)
    let mut r = &bytes[..];
    let mut e = || Fr::deserialize_compressed(&mut r).unwrap();
    let (eval_a, eval_b, eval_c, eval_q_add, eval_q_mul, eval_s_id, eval_s_sigma, eval_z, eval_t) = (e(), e(), e(), e(), e(), e(), e(), e(), e());
    let mut cm = || -> KZGCommitment { marlin_pc::Commitment::<Bn254>::deserialize_compressed(&mut r).unwrap() };
    let (comm_a, comm_b, comm_c, comm_q_add, comm_q_mul, comm_s_id, comm_s_sigma, comm_z, comm_t) = (cm(), cm(), cm(), cm(), cm(), cm(), cm(), cm(), cm());
    let mut op = || -> KZGOpeningProof { kzg10::Proof::<Bn254>::deserialize_compressed(&mut r).unwrap() };
    let (proof_a, proof_b, proof_c, proof_q_add, proof_q_mul, proof_s_id, proof_s_sigma, proof_z, proof_t) = (op(), op(), op(), op(), op(), op(), op(), op(), op());
    PlonkProof {
        eval_a, eval_b, eval_c, eval_q_add, eval_q_mul, eval_s_id, eval_s_sigma, eval_z, eval_t,
        comm_a, comm_b, comm_c, comm_q_add, comm_q_mul, comm_s_id, comm_s_sigma, comm_z, comm_t,
        proof_a, proof_b, proof_c, proof_q_add, proof_q_mul, proof_s_id, proof_s_sigma, proof_z, proof_t,
    }
}
