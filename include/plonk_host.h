/* libplonk_host: the prover of sabrinahirani/PLONK above libplonk_hip, as one call per proof.
 *
 * Plain C ABI of plonk_amd/host/ (C++ mirror of src/prover.rs, src/circuit.rs, src/poly_utils.rs, src/transcript.rs).  This is the
 * surface a drop-in `create_plonk_proof_with_transcript` binds (bindings/rust/src/prover_fast.rs): the whole five-round flow of
 * src/prover.rs:119-336 runs behind plonk_prove_host with the polynomials resident in HBM, the commitments and openings of a round in
 * flight together, and the host arrays crossing PCIe while the first commitments are computed.  The caller keeps what the reference
 * keeps on the CPU side of its own interfaces: the rng (it hands over the 27 field elements the prover would draw) and the
 * Poseidon opening challenge.
 *
 * Conventions: as include/plonk_hip.h (field elements = 4 little-endian uint64_t limbs in Montgomery form, points = x || y,
 * status int32_t with 0 = success and the PLK_ERR_* values of plonk_hip.h; nothing throws across the boundary).  After a non-zero
 * status plonk_host_last_error() returns the message the reference would have panicked with ("t(X) not divisible by Z_H(X)",
 * "index out of bounds: ...", "called `Option::unwrap()` on a `None` value: ...").  One thread per key, as in the reference.
 */
#ifndef PLONK_HOST_H
#define PLONK_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* plonk_host_last_error(void);

/* ---- KZGProverKey (src/prover.rs:32): the committer key of PCS::trim (src/main.rs:183) kept in HBM.
 * plonk_pk_load takes powers_of_g[0..len) and powers_of_gamma_g[0..3) as affine x || y (the fields `powers` and `powers_of_gamma_g`
 * of marlin_pc::CommitterKey); plonk_pk_ctx hands out the plk_ctx underneath for callers that also use plonk_hip.h directly. */
int32_t plonk_pk_create(int32_t device, void** pk_out);
int32_t plonk_pk_destroy(void* pk);
void* plonk_pk_ctx(void* pk);
int32_t plonk_pk_load(void* pk, const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy);
/* synthetic keys with a known trapdoor (tests, benchmarks): this rank's contiguous range / block-cyclic share of tau^i G */
int32_t plonk_pk_generate(void* pk, const uint64_t* tau, const uint64_t* gscal, uint64_t first, uint64_t count, uint64_t len);
int32_t plonk_pk_generate_cyclic(void* pk, const uint64_t* tau, const uint64_t* gscal, uint32_t rank, uint32_t world, uint32_t log_block, uint64_t len);
/* PCS::setup + PCS::trim (src/main.rs:182-183) from the reference's own rng (plonk_rng_create); vk48: g, gamma_g, h, beta_h */
int32_t plonk_pk_setup(void* pk, uint64_t max_degree, void* rng, uint32_t rank, uint32_t world, uint32_t log_block, uint64_t* vk48);
/* the 128-bit value every PCS::open squeezes from test_sponge() (src/prover.rs:36-59,285-303), as an Fr */
int32_t plonk_pk_set_opening_challenge(void* pk, const uint64_t* challenge);
/* multi-GPU: this process holds one shard of the key; gather(user, mine[count][16], count, all[world][count][16]) exchanges the
 * partial sums of a round (three calls per proof). */
typedef void (*plonk_gather_fn)(void* user, const uint64_t* mine, uint64_t count, uint64_t* all);
int32_t plonk_pk_set_sharding(void* pk, uint32_t rank, uint32_t world, plonk_gather_fn fn, void* user);
/* the same with the exchange inside the library (ncclAllGather over RCCL / xGMI, no callback): rank 0 obtains 128 bytes from
 * plonk_comm_get_unique_id and hands them to every rank by any channel; every rank calls plonk_pk_comm_init */
int32_t plonk_comm_get_unique_id(uint8_t* id128);
int32_t plonk_pk_comm_init(void* pk, const uint8_t* id128, uint32_t rank, uint32_t world);

/* ---- create_plonk_proof_with_transcript (src/prover.rs:119-336; caller src/main.rs:211-221).
 * polys: a, b, c, q_add, q_mul, s_id, s_sigma (coefficient arrays, trailing zeros are stripped as DensePolynomial::from_coefficients_vec
 * does), lens their lengths; witness_flat 3n elements, sigma 3n indices, n = 2^log_n; public_inputs as absorbed by src/prover.rs:138;
 * blinds: the Fr::rand draws of the nine commitments in the reference's order (27 elements: a, b, c, q_add, q_mul, s_id, s_sigma, z, t,
 * three each).  proof_bytes: 1170 bytes = the 27 fields of PlonkProof (src/prover.rs:61-95) in declaration order, ark-serialize
 * compressed: 9 x Fr (32 B), 9 x marlin_pc::Commitment (33 B), 9 x kzg10::Proof (65 B).  challenges (optional): beta, gamma, alpha,
 * zeta.  seconds (optional): 6 doubles of host wall clock per round.
 * plonk_prove_host borrows host arrays for the duration of the call (the reference's call shape); plonk_inputs_create / plonk_prove
 * split it into "make the inputs resident once" and "prove" for callers that prove the same circuit repeatedly. */
int32_t plonk_prove_host(void* pk, const uint64_t* const* polys, const uint64_t* lens, const uint64_t* witness_flat, const uint64_t* sigma, uint32_t log_n,
                         const uint64_t* public_inputs, uint64_t n_public, const uint64_t* blinds, uint64_t n_blinds, uint8_t* proof_bytes, uint64_t* challenges,
                         double* seconds);
int32_t plonk_inputs_create(void* pk, const uint64_t* const* polys, const uint64_t* lens, const uint64_t* witness_flat, const uint64_t* sigma, uint32_t log_n,
                            void** inputs_out);
int32_t plonk_inputs_destroy(void* inputs);
int32_t plonk_prove(void* pk, void* inputs, const uint64_t* public_inputs, uint64_t n_public, const uint64_t* blinds, uint64_t n_blinds, uint8_t* proof_bytes,
                    uint64_t* challenges, double* seconds);
/* the same with the blinding coefficients drawn from a restated rand 0.8 StdRng (`rng: &mut R` of src/prover.rs:132) */
int32_t plonk_prove_rng(void* pk, void* inputs, const uint64_t* public_inputs, uint64_t n_public, void* rng, uint8_t* proof_bytes, uint64_t* challenges,
                        double* seconds);

/* ---- verify_plonk_proof_with_transcript (src/verifier.rs:57-236) on the CPU, for hosts without arkworks: *ok = 1 when all nine
 * openings check.  plonk_verify builds the verifier key of a synthetic SRS from its trapdoor, plonk_verify_vk takes one (vk48). */
int32_t plonk_verify(const uint64_t* tau, const uint64_t* gscal, const uint8_t* proof_bytes, uint64_t proof_len, const uint64_t* public_inputs, uint64_t n_public,
                     const uint64_t* opening_challenge, int32_t* ok);
int32_t plonk_verify_vk(const uint64_t* vk48, const uint8_t* proof_bytes, uint64_t proof_len, const uint64_t* public_inputs, uint64_t n_public,
                        const uint64_t* opening_challenge, int32_t* ok);

/* ---- the driver's randomness for hosts without ark-std (src/main.rs:25, src/prover.rs:48): test_rng() when seed32 is NULL */
int32_t plonk_rng_create(const uint8_t* seed32, void** rng_out);
int32_t plonk_rng_destroy(void* rng);
uint64_t plonk_rng_next_u64(void* rng);
uint32_t plonk_rng_next_u32(void* rng);
void plonk_rng_fr(void* rng, uint64_t n, uint64_t* out);
void plonk_rng_g1(void* rng, uint64_t* xy);
void plonk_rng_g2(void* rng, uint64_t* x8y8);
void plonk_test_sponge_challenge(uint64_t* out4);

#ifdef __cplusplus
}
#endif
#endif
