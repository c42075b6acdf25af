/* libplonk_hip: MI355X (gfx950) backend for the prover hot path of sabrinahirani/PLONK.
 *
 * Plain C ABI.  The reference has no FFI of its own: each entry point below names the reference
 * function (or third-party call made by the reference) whose arithmetic it replaces.  RNG, the
 * Poseidon opening challenge, the merlin transcript and serialization stay in the caller.
 *
 * Conventions
 *   - Field elements: 4 little-endian uint64_t limbs in Montgomery form (x * 2^256 mod p), i.e. the
 *     in-memory payload of arkworks' Fp256.  Fr = BN254 scalar field, Fq = base field.
 *   - G1 points: x || y (8 limbs) plus a separate infinity flag.  Never a Rust struct by pointer.
 *   - Indices: uint64_t.  Status: int32_t, 0 = PLK_OK.  Nothing aborts or throws across the ABI.
 *   - "host" pointers are ordinary memory owned by the caller; "dev" pointers are device memory
 *     obtained from plk_malloc (or any hipMalloc'd memory of the same device).
 *   - Calls on one plk_ctx must be serialised by the caller (the reference is single threaded).
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct plk_ctx plk_ctx;

enum {
    PLK_OK = 0,
    PLK_ERR_INVALID_ARG = 1,
    PLK_ERR_HIP = 2,              /* a HIP runtime call failed; see plk_last_error */
    PLK_ERR_NO_DEVICE = 3,
    PLK_ERR_SRS_TOO_SHORT = 4,    /* arkworks: Error::TooManyCoefficients (unwrap panics, src/prover.rs:150) */
    PLK_ERR_NOT_DIVISIBLE = 5,    /* assert!(remainder.is_zero(), "t(X) not divisible by Z_H(X)") src/circuit.rs:487 */
    PLK_ERR_INVERSE_OF_ZERO = 6,  /* .inverse().unwrap() on zero, src/circuit.rs:377,523 */
    PLK_ERR_INDEX_OUT_OF_RANGE = 7, /* sigma entry >= 3n (index panic in src/circuit.rs:508-510) */
    PLK_ERR_ALLOC = 8
};

/* ---- lifecycle -------------------------------------------------------------------------- */
int32_t plk_create(int32_t device, plk_ctx** out);
int32_t plk_destroy(plk_ctx* ctx);
const char* plk_last_error(plk_ctx* ctx);

/* ---- device memory ---------------------------------------------------------------------- */
int32_t plk_malloc(plk_ctx* ctx, uint64_t bytes, void** dev);
int32_t plk_free(plk_ctx* ctx, void* dev);
int32_t plk_upload(plk_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes);
int32_t plk_download(plk_ctx* ctx, void* host_dst, const void* dev_src, uint64_t bytes);
/* Upload without waiting, for a host that hands over its polynomials with the call (src/prover.rs:119-133 borrows host vectors): the
 * transfer is enqueued on a stream of its own and every LATER call on this context that reads dev_dst is ordered after it, so the
 * commitment of the first polynomial runs while the others are still crossing PCIe.  host_src must stay valid until a call that
 * returns results of work reading dev_dst has returned (or plk_sync).  From pinned memory (plk_host_malloc) the call returns at
 * once and the transfer runs at link speed; ordinary memory (a Rust Vec, a std::vector) goes through a ring of page-locked buffers
 * filled by helper threads of the library, at link speed as well, and the call returns when the array has been copied out. */
int32_t plk_upload_enqueue(plk_ctx* ctx, void* dev_dst, const void* host_src, uint64_t bytes);
/* page-locked host memory for plk_upload_enqueue */
int32_t plk_host_malloc(plk_ctx* ctx, uint64_t bytes, void** host);
int32_t plk_host_free(plk_ctx* ctx, void* host);
int32_t plk_sync(plk_ctx* ctx);
/* device-to-device copy of `bytes`, then zero fill up to `total_bytes` (zero padding of DensePolynomial coefficients) */
int32_t plk_copy_dev(plk_ctx* ctx, void* dev_dst, const void* dev_src, uint64_t bytes, uint64_t total_bytes);

/* ---- structured reference string (KZGProverKey, src/prover.rs:32; built by PCS::trim, src/main.rs:183)
 * powers_of_g[0..len) and powers_of_gamma_g[0..3) as affine x||y.  The library keeps them resident in
 * HBM together with its window tables (2^(17 j) * P for each window j; 16- and 13-bit tables over a
 * prefix of the points).  With n_shards > 1 this context
 * keeps only powers [shard_first, shard_first + shard_len) and every MSM it runs is the partial sum over
 * that range; the gamma powers belong to the shard that contains index 0. */
int32_t plk_srs_load(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t len, const uint64_t* gamma_g_xy_host);
int32_t plk_srs_load_shard(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t shard_first, uint64_t shard_len,
                           uint64_t total_len, const uint64_t* gamma_g_xy_host);
/* Synthetic SRS with a known trapdoor (what PCS::setup draws from its rng, src/main.rs:182):
 * powers_of_g[i] = tau^i G, powers_of_gamma_g[j] = gscal tau^j G, computed on the device. */
int32_t plk_srs_generate(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint64_t len);
int32_t plk_srs_generate_shard(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint64_t shard_first,
                               uint64_t shard_len, uint64_t total_len);
/* Block-cyclic shards: the powers are cut into blocks of 2^log_block points and block b belongs to rank b mod world, so that every
 * MSM, whatever its length, is spread evenly over the ranks (with contiguous ranges of a 4n-point SRS the n-coefficient polynomials
 * of the prover would all land on the first ranks).  plk_srs_load_cyclic takes the whole array of powers and keeps this rank's blocks.
 * Everything else is as for contiguous shards: MSMs return this rank's partial sum, the gamma powers live on rank 0. */
int32_t plk_srs_generate_cyclic(plk_ctx* ctx, const uint64_t tau[4], const uint64_t gscal[4], uint32_t rank, uint32_t world,
                                uint32_t log_block, uint64_t total_len);
int32_t plk_srs_load_cyclic(plk_ctx* ctx, const uint64_t* g_xy_host, uint64_t total_len, const uint64_t* gamma_g_xy_host, uint32_t rank,
                            uint32_t world, uint32_t log_block);
/* PCS::setup + PCS::trim (src/main.rs:182-183; ark-poly-commit KZG10::setup) from the three values setup draws from its rng:
 * powers_of_g[i] = beta^i g for i < len and powers_of_gamma_g[j] = beta^j gamma_g for j < 3, computed on the device through
 * fixed-base window tables of g and gamma_g (arkworks' g.batch_mul(powers_of_beta)).  g and gamma_g: affine x||y, any points of the
 * curve.  The cyclic form keeps this rank's blocks as plk_srs_generate_cyclic does (world <= 1: everything). */
int32_t plk_srs_setup(plk_ctx* ctx, const uint64_t beta[4], const uint64_t g_xy[8], const uint64_t gamma_g_xy[8], uint64_t len);
int32_t plk_srs_setup_cyclic(plk_ctx* ctx, const uint64_t beta[4], const uint64_t g_xy[8], const uint64_t gamma_g_xy[8], uint32_t rank,
                             uint32_t world, uint32_t log_block, uint64_t total_len);
/* Reads points back: which = 0 powers_of_g (global index first..first+count), 1 powers_of_gamma_g. */
int32_t plk_srs_read(plk_ctx* ctx, int32_t which, uint64_t first, uint64_t count, uint64_t* out_xy_host);

/* ---- variable-base MSM (ark-ec VariableBaseMSM::msm_bigint, reached from PCS::commit / PCS::open,
 * src/prover.rs:150-156,210,253,295-303):  sum_i scalars[i] * powers_of_g[base_offset + i]. */
int32_t plk_msm(plk_ctx* ctx, const uint64_t* scalars_host, uint64_t k, uint64_t base_offset, uint64_t out_xy[8],
                int32_t* out_is_inf);
int32_t plk_msm_dev(plk_ctx* ctx, const void* scalars_dev, uint64_t k, uint64_t base_offset, uint64_t out_xy[8],
                    int32_t* out_is_inf);
/* Partial result of this context's shard as an unnormalised point X,Y,ZZ,ZZZ (x = X/ZZ, y = Y/ZZZ;
 * ZZ = 0 is the identity), 16 limbs.  plk_g1_sum_partials adds such partials and normalises. */
int32_t plk_msm_partial_dev(plk_ctx* ctx, const void* scalars_dev, uint64_t k, uint64_t base_offset, uint64_t out_xyzz[16]);
int32_t plk_g1_sum_partials(const uint64_t* xyzz, uint64_t count, uint64_t out_xy[8], int32_t* out_is_inf);
/* The window layout this context would use for the MSM over [base_offset, base_offset + k) (no counterpart in the reference, whose
 * msm_bigint picks its window width inside ark-ec; for benchmarks and tests that account for the work of an MSM): bits per window,
 * windows per scalar (= mixed additions per scalar on this rank), buckets. */
int32_t plk_msm_plan_query(plk_ctx* ctx, uint64_t k, uint64_t base_offset, int32_t* window_bits, int32_t* windows, int32_t* buckets);

/* ---- radix-2 NTT (ark-poly Radix2EvaluationDomain::fft / ifft; src/prover.rs:232,233,237,
 * src/circuit.rs:404,438,454,543, src/poly_utils.rs:9,27,28).  In place, natural order in and out, size 2^log_n.
 * inverse != 0 multiplies by n^-1.  coset_gen != NULL: forward evaluates on g*w^i, inverse interpolates from it. */
int32_t plk_ntt(plk_ctx* ctx, uint64_t* data_host, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen);
int32_t plk_ntt_dev(plk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen);
/* the same without waiting: the transform is ordered before every later call on this context (plk_sync or any call that returns data waits) */
int32_t plk_ntt_dev_enqueue(plk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen);

/* ---- building blocks of the multi-GPU (four-step) NTT: N = G * M points over G <= 8 devices, one all-to-all per transform
 * (plonk_amd/dist_ntt.py drives them).  The reference has no counterpart: its transforms are single ark-poly fft / ifft calls
 * (src/poly_utils.rs:27-28, src/prover.rs:232-237); these split one such transform across devices.
 * plk_scale_powers_dev:  data[i] *= base * ratio^i, i < n  (the w_N^(g k2) twiddles between the two steps, coset factors).
 * plk_ntt_cross_dev:     data is a [2^log_g][cols] array; every column is replaced by its 2^log_g-point transform
 *                        (row k1 <- sum_g w_G^(+-g k1) row g; no 1/G factor in either direction), log_g <= 3.
 * Both are enqueued on the context's stream like plk_ntt_dev_enqueue. */
int32_t plk_scale_powers_dev(plk_ctx* ctx, void* data_dev, uint64_t n, const uint64_t base[4], const uint64_t ratio[4]);
int32_t plk_ntt_cross_dev(plk_ctx* ctx, void* data_dev, uint32_t log_g, uint64_t cols, int32_t inverse);

/* ---- interpolation helpers (src/poly_utils.rs:7-11 and :14-31).  sigma has sigma_len <= 2^log_n4 entries and is
 * padded with the identity; outputs are 2^log_n4 coefficients each (the caller strips trailing zeros, as
 * DensePolynomial::from_coefficients_vec does). */
int32_t plk_interpolate_selector(plk_ctx* ctx, const uint64_t* values_host, uint64_t len, uint64_t* coeffs_out_host, uint32_t* log_n_out);
int32_t plk_interpolate_permutation(plk_ctx* ctx, const uint64_t* sigma_host, uint64_t sigma_len, uint32_t log_n4,
                                    uint64_t* s_id_out_host, uint64_t* s_sigma_out_host);
int32_t plk_interpolate_permutation_dev(plk_ctx* ctx, const void* sigma_dev, uint64_t sigma_len, uint32_t log_n4,
                                        void* s_id_out_dev, void* s_sigma_out_dev);

/* ---- Circuit::build_grand_product (src/circuit.rs:491-544) with s_id_vals[j] = j (src/prover.rs:205).
 * witness_flat: 3n Fr ([A_i,B_i,C_i]); sigma: 3n indices; out: n coefficients of ifft(z[0..n)). */
int32_t plk_grand_product(plk_ctx* ctx, const uint64_t* witness_flat_host, const uint64_t* sigma_host, uint32_t log_n,
                          const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_coeffs_out_host);
int32_t plk_grand_product_dev(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n,
                              const uint64_t beta[4], const uint64_t gamma[4], void* z_coeffs_out_dev);
/* The same, also handing out the values z[0..n) on H that the interpolation starts from: the prover's
 * z_vals = domain.fft(&z_poly.coeffs) (src/prover.rs:237) are exactly these, so the forward transform can be skipped. */
int32_t plk_grand_product_values_dev(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n,
                                     const uint64_t beta[4], const uint64_t gamma[4], void* z_vals_out_dev, void* z_coeffs_out_dev);
/* The same without the final wait: the errors the call can report are known when it returns, the last kernels are still running on
 * the compute stream.  Commitments, plk_compute_detach and compute calls issued afterwards are ordered behind them; a host read of
 * the outputs needs plk_sync first.  z_coeffs_out_dev may be NULL: values only, no interpolation (the caller interpolates a copy
 * with plk_ntt_dev_enqueue).  Lets the prover start round 3 (src/prover.rs:224-259), which reads the values, beside the
 * interpolation and the sort of z (:208-216). */
int32_t plk_grand_product_dev_enqueue(plk_ctx* ctx, const void* witness_flat_dev, const void* sigma_dev, uint32_t log_n,
                                      const uint64_t beta[4], const uint64_t gamma[4], void* z_vals_out_dev, void* z_coeffs_out_dev);

/* ---- Circuit::build_quotient_polynomial (src/circuit.rs:442-489) incl. build_permutation_constraint (:343-405)
 * and the all-zero public-input polynomial of src/prover.rs:224-227.  Columns are values on H (n each).
 * Writes the coefficients of t into t_out (n entries, zero filled) and their count (0 for the zero polynomial).
 * Returns PLK_ERR_NOT_DIVISIBLE when the numerator is not a multiple of X^n - 1. */
int32_t plk_quotient(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals,
                     const uint64_t* q_mul_vals, const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n,
                     const uint64_t beta[4], const uint64_t gamma[4], uint64_t* t_out_host, uint64_t* t_len);
int32_t plk_quotient_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals,
                         const void* q_mul_vals, const void* z_vals, const void* sigma, uint32_t log_n,
                         const uint64_t beta[4], const uint64_t gamma[4], void* t_out_dev, uint64_t* t_len);

/* The same with the public-input polynomial of Circuit::build_public_input_poly (src/circuit.rs:410-439), which the prover's own
 * temporary circuit never has (src/prover.rs:224-227) but a caller of build_quotient_polynomial on a circuit with marked public
 * inputs does:  constraint[pi_rows[k]] = alpha * (a[pi_rows[k]] - pi_values[k])  for k = 0 .. pi_count - 1 in that order (a later
 * entry for the same row replaces an earlier one, as the reference's assignment does), zero elsewhere; its interpolant is added to
 * the numerator.  pi_rows[k] is the first gate row in which public variable k appears as an input or output (src/circuit.rs:424-429;
 * resolved by the caller, who holds the builder; a variable that appears in no gate is left out, as the reference skips it), pi_values[k]
 * its value.  pi_rows / pi_values are host arrays in both forms.  A row >= n gives PLK_ERR_INDEX_OUT_OF_RANGE (the reference would index
 * out of bounds).  Since a[row] is the row's first wire, the term vanishes only when that wire carries the public value; otherwise the
 * call returns PLK_ERR_NOT_DIVISIBLE exactly where the reference's assert fires.  pi_count = 0 is plk_quotient. */
int32_t plk_quotient_pi(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals,
                        const uint64_t* q_mul_vals, const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n,
                        const uint64_t beta[4], const uint64_t gamma[4], const uint64_t alpha[4], const uint64_t* pi_rows,
                        const uint64_t* pi_values, uint64_t pi_count, uint64_t* t_out_host, uint64_t* t_len);
int32_t plk_quotient_pi_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals,
                            const void* q_mul_vals, const void* z_vals, const void* sigma, uint32_t log_n,
                            const uint64_t beta[4], const uint64_t gamma[4], const uint64_t alpha[4], const uint64_t* pi_rows_host,
                            const uint64_t* pi_values_host, uint64_t pi_count, void* t_out_dev, uint64_t* t_len);

/* ---- quotient on the extended coset.  No counterpart in the reference, which evaluates its constraints on H only and therefore
 * always obtains t = 0 (src/circuit.rs:442-489); this is the same constraint system taken as polynomial identities, the form a
 * prover with a non-trivial quotient needs.  With a, b, c, q_add, q_mul, z the degree < n interpolants of the given values on H,
 * ID_k the interpolant of the labels 3i + k (src/prover.rs:205) and SG_k of sigma(3i + k) (src/circuit.rs:356-368), k < 3:
 *   num(X) = q_add (a + b - c) + q_mul (a b - c)                                     (gate, src/circuit.rs:446-453)
 *          + alpha [ z(X) prod_k (w_k + beta ID_k + gamma) - z(wX) prod_k (w_k + beta SG_k + gamma) ]   (src/circuit.rs:343-379)
 *   t(X)   = num(X) / (X^n - 1)
 * evaluated on the coset g H_4n (g = coset_gen, NULL = 5) and interpolated back: t_out receives 4n coefficients (the last n + 3
 * are zero).  Returns PLK_ERR_NOT_DIVISIBLE when num is not a multiple of X^n - 1 (t_out then holds the coset interpolant). */
int32_t plk_quotient_coset(plk_ctx* ctx, const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* q_add_vals,
                           const uint64_t* q_mul_vals, const uint64_t* z_vals, const uint64_t* sigma, uint32_t log_n, const uint64_t beta[4],
                           const uint64_t gamma[4], const uint64_t alpha[4], const uint64_t* coset_gen, uint64_t* t_out_host);
int32_t plk_quotient_coset_dev(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals, const void* q_mul_vals,
                               const void* z_vals, const void* sigma, uint32_t log_n, const uint64_t beta[4], const uint64_t gamma[4],
                               const uint64_t alpha[4], const uint64_t* coset_gen, void* t_out_dev);

/* ---- DensePolynomial::evaluate (src/prover.rs:266-274) */
int32_t plk_poly_eval(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t x[4], uint64_t out[4]);
int32_t plk_poly_eval_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t x[4], uint64_t out[4]);

/* The same evaluation for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  The block sums and carries the kernels exchange live in a fixed array
 * in device memory (524,608 bytes per GPU, held until the process ends) instead of the context's grow-on-demand buffer: the call
 * performs no hipMalloc / hipFree, at any length up to PLK_POLY_EVAL_STATIC_MAX_LEN coefficients (longer: PLK_ERR_INVALID_ARG).
 * out_dev: 32 bytes of device memory that receive the value.  The work is enqueued on the context's stream and not waited for:
 * it is ordered before every later call on this context (plk_sync or plk_download waits).  All contexts of one GPU share the
 * array: calls that overlap on one GPU race. */
#define PLK_POLY_EVAL_STATIC_MAX_LEN (1ull << 26)
int32_t plk_poly_eval_dev_static(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t x[4], void* out_dev);

/* ---- PCS::commit with hiding bound 1 (ark-poly-commit kzg10::commit; src/prover.rs:150-156,210,253):
 * C = sum coeffs[i] powers_of_g[i] + sum_{j<3} blind[j] powers_of_gamma_g[j].  blind = 12 limbs. */
int32_t plk_kzg_commit(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t* blind, uint64_t out_xy[8], int32_t* out_is_inf);
int32_t plk_kzg_commit_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, uint64_t out_xy[8], int32_t* out_is_inf);
/* ---- PCS::open for one polynomial (marlin_pc::open + kzg10::open; src/prover.rs:295-303): opens challenge*f with
 * randomness challenge*r at zeta.  W = MSM((challenge f)/(X - zeta)) + MSM_gamma((challenge r)/(X - zeta)),
 * random_v = (challenge r)(zeta), eval = f(zeta) (the division's remainder divided by challenge). */
int32_t plk_kzg_open(plk_ctx* ctx, const uint64_t* coeffs_host, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                     const uint64_t challenge[4], uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]);
int32_t plk_kzg_open_dev(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                         const uint64_t challenge[4], uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]);

/* Split forms of the two calls above: *_begin enqueues the device work on one of the context's streams and
 * returns a ticket; *_end waits for it and returns the result.  Several tickets may be open at once, which is how
 * the seven round-1 commitments and the nine openings of src/prover.rs overlap their latency-bound tails. */
int32_t plk_kzg_commit_begin(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, int32_t* ticket);
int32_t plk_kzg_commit_end(plk_ctx* ctx, int32_t ticket, uint64_t out_xy[8], int32_t* out_is_inf);
int32_t plk_kzg_open_begin(plk_ctx* ctx, const void* coeffs_dev, uint64_t len, const uint64_t* blind, const uint64_t zeta[4],
                           const uint64_t challenge[4], int32_t* ticket);
int32_t plk_kzg_open_end(plk_ctx* ctx, int32_t ticket, uint64_t w_xy[8], int32_t* w_is_inf, uint64_t random_v[4], uint64_t eval[4]);
/* plk_kzg_open_begin for `count` polynomials opened at one point with one challenge (the nine openings of src/prover.rs:295-303):
 * coeffs_dev[k] / len[k] / blind + 12 k describe polynomial k, tickets[k] receives its ticket; each is ended with plk_kzg_open_end or
 * plk_kzg_end_partial like any other.  Same results as `count` calls of plk_kzg_open_begin in the same order; the divisions by
 * (X - zeta) of all polynomials run as one batch of launches before the first sort.  count <= 10; all tickets of one batch must have
 * ended before the next batch begins. */
int32_t plk_kzg_open_begin_batch(plk_ctx* ctx, uint32_t count, const void* const* coeffs_dev, const uint64_t* len, const uint64_t* blind,
                                 const uint64_t zeta[4], const uint64_t challenge[4], int32_t* tickets);
/* plk_kzg_commit_begin for the `count` commitments of a round (src/prover.rs:150-156), same conventions as the batch of openings. */
int32_t plk_kzg_commit_begin_batch(plk_ctx* ctx, uint32_t count, const void* const* coeffs_dev, const uint64_t* len, const uint64_t* blind,
                                   int32_t* tickets);
/* Ordering of a *_begin call against later calls on the same context: the polynomial a *_begin call reads (coeffs_dev) may be
 * overwritten by any later compute call of this context (plk_ntt_dev_enqueue in place, plk_copy_dev, plk_upload, a quotient or
 * grand product writing into it, ...): the library orders that call behind the ticket's last read.  Two exceptions, both by
 * contract: compute work issued while detached (plk_compute_detach) must not write what an open ticket still reads, and a
 * plk_upload_enqueue after the first one of a batch (the batch ends with plk_kzg_submit) must not target a buffer that a *_begin
 * call made since that first upload reads. */
/* Tells the library that the batch of tickets begun so far is complete.  The last stage of every MSM (bucket reduction and
 * read-back) shares one stream; it is enqueued for the whole batch at once, ordered by expected completion, either here or at
 * the first *_end of the batch.  Calling this right after the last *_begin lets that stage start while the host does other work. */
int32_t plk_kzg_submit(plk_ctx* ctx);
/* *ready = 1 when the ticket's result has arrived and its *_end will not wait.  A host with several tickets open ends the ready ones
 * first: each *_end finishes its MSM with about a hundred point additions on the CPU, which then run while the GPU is still busy with
 * the others instead of after the last one. */
int32_t plk_kzg_ready(plk_ctx* ctx, int32_t ticket, int32_t* ready);
/* Gives up every open ticket: waits for the work they enqueued, discards the results and frees their slots.  No counterpart in the
 * reference, where a panic unwinds the whole proof; a host that turns a failed status into an error and carries on calls this so that
 * the tickets begun before the failure do not stay open. */
int32_t plk_kzg_cancel(plk_ctx* ctx);
/* Like *_end but returns this shard's unnormalised partial (16 limbs) instead of the affine point. */
int32_t plk_kzg_end_partial(plk_ctx* ctx, int32_t ticket, uint64_t out_xyzz[16], uint64_t random_v[4], uint64_t eval[4]);

/* Ends `count` tickets with one call.  out: 16 limbs per ticket, the affine point x || y in the first eight when affine != 0 (and
 * is_inf[k]), the unnormalised partial sum (as plk_kzg_end_partial) otherwise; random_v / eval: 4 limbs per ticket, written for
 * openings, may be NULL.  Same values as the single calls; the host-side ends of the MSMs (about 140 point additions and one
 * inversion each) are shared among a few threads of the library, and results that have arrived are finished while the device still
 * works on the others.  plk_g1_sum_partials_many is plk_g1_sum_partials for `count` points: xyzz = [world][count][16] limbs. */
int32_t plk_kzg_end_many(plk_ctx* ctx, uint32_t count, const int32_t* tickets, int32_t affine, uint64_t* out, int32_t* is_inf, uint64_t* random_v,
                         uint64_t* eval);
int32_t plk_g1_sum_partials_many(const uint64_t* xyzz, uint64_t world, uint64_t count, uint64_t* out_xy, int32_t* out_is_inf);

/* ---- multi-GPU exchange inside the library (no counterpart in the reference, which is a single process).  One process per GPU;
 * every rank holds a shard of the SRS (plk_srs_*_cyclic), runs the same prover, and every commitment or opening is the sum of the
 * ranks' partial sums (plk_kzg_end_partial).  Rank 0 calls plk_comm_get_unique_id and hands the 128 bytes to the other ranks by any
 * channel (a file, MPI, a torch.distributed broadcast); every rank then calls plk_comm_init (ncclCommInitRank on this context's
 * device; RCCL is opened at run time, so single-GPU users need no librccl).  plk_comm_all_gather_partials exchanges `count`
 * partial sums (16 limbs each) with one ncclAllGather on the context's stream: mine[count][16] -> all[world][count][16], host
 * arrays; plk_g1_sum_partials adds what it returns (point addition is not a reduction RCCL offers: gather, then add). */
int32_t plk_comm_get_unique_id(uint8_t id[128]);
int32_t plk_comm_init(plk_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world);
int32_t plk_comm_destroy(plk_ctx* ctx);
int32_t plk_comm_all_gather_partials(plk_ctx* ctx, const uint64_t* mine, uint64_t count, uint64_t* all);

/* Four-step NTT of N = 2^log_n points over the G = 1, 2, 4 or 8 ranks of the communicator, one all-to-all per transform, everything
 * enqueued on the context's stream (no device-wide wait, no host round trip): the multi-GPU form of the ark-poly fft / ifft calls
 * the reference makes at src/poly_utils.rs:27-28 and src/prover.rs:232-237, for transforms that are worth spreading (2^24 points and
 * up).  Every rank passes its M = N / G elements in data_dev and a second buffer of M elements in tmp_dev.  One exchange means
 * different layouts on the two sides (like the bit-reversed output of an in-place FFT):
 *   coefficients  cyclic:         rank g holds c[g + G m] at m                          (input of forward, output of inverse)
 *   evaluations   column blocks:  rank r holds X[(r M/G + k2) + M k1] at k1 M/G + k2    (output of forward, input of inverse)
 * forward: local (coset) NTT of M points -> times h^g w_N^(g k2) -> all-to-all (grouped ncclSend / ncclRecv, one block of M / G
 * elements per pair of ranks, each over its own xGMI link) -> G-point transforms across the received blocks; inverse: the same
 * backwards, 1/N folded into the twiddle step and the local inverse NTT.  coset_gen as for plk_ntt (NULL: none).
 * *result_in_tmp tells which buffer holds the result (tmp_dev whenever an exchange took place); the other one is scratch.
 * phase 0: the whole transform (rank and world are the communicator's; the two arguments are ignored).
 * phase 1 / 2: only this rank's work before / after the exchange, in place in data_dev, for `rank` of `world` as given and without
 *   a communicator: for callers that move the blocks themselves (logical ranks on one device in the tests, another transport).
 * phase 3: phase 0, with the exchange performed even when the communicator has a single rank (it sends to itself). */
int32_t plk_ntt_sharded_dev(plk_ctx* ctx, void* data_dev, void* tmp_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen, uint32_t phase,
                            uint32_t rank, uint32_t world, int32_t* result_in_tmp);

/* ---- scheduling hint with no counterpart in the reference (which runs everything in sequence).  The library runs NTT, pointwise and
 * scan kernels ("compute" work) on the same stream as the sorting stage of its MSMs, so that a commitment sees the polynomial the
 * call before it produced.  Between plk_compute_detach(ctx, 1) and plk_compute_detach(ctx, 0) compute work goes to a second stream
 * and runs beside the sorts and accumulations of the *_begin calls made in the same span: the selector transforms of
 * src/prover.rs:232-233 beside the first commitment's sort, build_quotient_polynomial (src/prover.rs:248) beside z's.  Detaching orders
 * the detached work after everything enqueued before; re-attaching orders everything after it.  In between the caller guarantees
 * that no *_begin call reads what a detached compute call writes.  on = 2 re-attaches WITHOUT ordering later calls after the
 * detached work (it keeps running beside them; only detached work enqueued later, or work after a later plk_compute_detach(ctx, 0),
 * may read its results).  on = 3 detaches WITHOUT ordering the detached work after what was enqueued since the last detach: for a
 * caller who detached once to fix the order (on = 1), re-attached (on = 2), enqueued attached work that the detached work does not
 * read, and now goes on with detached work.  Results never depend on this call, only timing does. */
int32_t plk_compute_detach(plk_ctx* ctx, int32_t on);
/* plk_quotient_dev in two halves, so that the host can enqueue other work while the quotient's kernels run:
 * _enqueue launches them (on the compute stream, detached or not) and returns; _finish waits and reports as plk_quotient_dev does. */
int32_t plk_quotient_dev_enqueue(plk_ctx* ctx, const void* a, const void* b, const void* c, const void* q_add_vals,
                                 const void* q_mul_vals, const void* z_vals, const void* sigma, uint32_t log_n,
                                 const uint64_t beta[4], const uint64_t gamma[4], void* t_out_dev);
int32_t plk_quotient_dev_finish(plk_ctx* ctx, uint64_t* t_len);

/* ---- element-wise field helpers used by tests and the microbenchmarks (op: 0 add, 1 sub, 2 mul; field: 0 Fr, 1 Fq) */
int32_t plk_field_op(plk_ctx* ctx, int32_t field, int32_t op, const uint64_t* a_host, const uint64_t* b_host, uint64_t* out_host, uint64_t n);

/* ---- measurement: accumulated device time (HIP events on the launching stream) and launch count of the named
 * kernel class since the last reset: "msm_accumulate", "msm_sort", "msm_reduce", "ntt", "pointwise", "division". */
/* on: 0 off, 1 every class, 2 "msm_accumulate" only (two events per MSM: cheap enough to leave on while a run is timed). */
int32_t plk_profile_enable(plk_ctx* ctx, int32_t on);
int32_t plk_profile_reset(plk_ctx* ctx);
int32_t plk_profile_get(plk_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);
/* Time during which at least one kernel of the class was running (launches of one class overlap when they sit on
 * different streams, so this can be less than the sum of their durations). */
int32_t plk_profile_get_busy(plk_ctx* ctx, const char* name, double* busy_ms);

#ifdef __cplusplus
}
#endif
#endif
