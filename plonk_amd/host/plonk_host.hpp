// Host side of the MI355X prover: the reference's Rust surface restated in C++ above the C ABI of
// include/plonk_hip.h.  Names, argument meaning and failure behaviour follow the reference:
//   src/prover.rs      PlonkProof, KZGCommitment, KZGOpeningProof, KZGProverKey, create_plonk_proof_with_transcript
//   src/circuit.rs     Circuit::build_grand_product, Circuit::build_quotient_polynomial
//   src/poly_utils.rs  interpolate_selector, interpolate_permutation_polynomials
//   src/transcript.rs  PlonkTranscript, challenge_scalar, absorb_*
// The reference panics on failure (unwrap / assert!); here every failure is a plonk::Error exception whose text is the
// reference's panic message where it has one.  Nothing in this directory touches the CPU oracle.
#pragma once
#include <stdint.h>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/plonk_hip.h"

namespace plonk {

struct Error : std::runtime_error {
    int32_t status;
    Error(int32_t s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// ark_bn254::Fr in arkworks' in-memory form: 4 little-endian limbs of x * 2^256 mod r
struct Fr {
    uint64_t v[4];
    bool operator==(const Fr& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
};
struct G1Affine {
    uint64_t xy[8];      // x || y, each x * 2^256 mod q
    int32_t infinity;
};
// marlin_pc::Commitment { comm, shifted_comm: None }
struct KZGCommitment {
    G1Affine comm;
};
// kzg10::Proof { w, random_v: Some(_) }
struct KZGOpeningProof {
    G1Affine w;
    Fr random_v;
};
// src/prover.rs:61-95, same field order
struct PlonkProof {
    Fr eval_a, eval_b, eval_c, eval_q_add, eval_q_mul, eval_s_id, eval_s_sigma, eval_z, eval_t;
    KZGCommitment comm_a, comm_b, comm_c, comm_q_add, comm_q_mul, comm_s_id, comm_s_sigma, comm_z, comm_t;
    KZGOpeningProof proof_a, proof_b, proof_c, proof_q_add, proof_q_mul, proof_s_id, proof_s_sigma, proof_z, proof_t;
};
// the four Fiat-Shamir challenges and per-round wall-clock seconds, for reports and tests
struct ProveTrace {
    Fr beta, gamma, alpha, zeta;
    double seconds[6];   // round-1 commitments, grand product (+commit), quotient (+commit), openings (+evaluations), assembly, total
};

// DensePolynomial<Fr>: coefficient vector without trailing zeros
struct DensePolynomial {
    std::vector<Fr> coeffs;
    static DensePolynomial from_coefficients_vec(std::vector<Fr> c) {
        while (!c.empty() && c.back().is_zero()) c.pop_back();
        return DensePolynomial{std::move(c)};
    }
};
// GeneralEvaluationDomain<Fr> of size 2^log_size
struct Domain {
    uint32_t log_size;
    uint64_t size() const { return 1ull << log_size; }
    static Domain with_at_least(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) ++l; return Domain{l}; }
};

// stands for `rng: &mut R`: every call is one Fr::rand(rng)
using Rng = std::function<Fr()>;

// ark_std::test_rng() (src/main.rs:25, src/prover.rs:48): rand 0.8 StdRng, which is ChaCha with 12 rounds read through a
// 64-word block buffer.  The deterministic branch of ark-std 0.5 (environment variable DETERMINISTIC_TEST_RNG=1).
class StdRng {
public:
    explicit StdRng(const uint8_t seed[32]);
    uint32_t next_u32();
    uint64_t next_u64();
    bool gen_bool();              // rng.gen::<bool>(): the sign bit of one u32
private:
    void refill(uint32_t index);
    uint8_t key_[32];
    uint64_t counter_ = 0;
    uint32_t buf_[64];
    uint32_t index_ = 64;
};
StdRng test_rng();
// one ChaCha block with rand_chacha's word layout (words 12-13 block counter, 14-15 stream id)
void chacha_block(const uint8_t key[32], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]);
Fr fr_rand(StdRng& rng);         // Fr::rand(rng)
Rng as_rng(StdRng& rng);         // the same generator behind the prover's `Rng`
struct G2Affine;
G1Affine g1_rand(StdRng& rng);   // G1Projective::rand(rng), affine
// the 128-bit challenge PCS::open squeezes from test_sponge() (src/prover.rs:36-59,285-303): the same value for all nine openings
Fr test_sponge_challenge();
struct KZGVerifierKey;

// Collective used when the SRS is sharded over several GPUs: gathers every rank's `count` partial points (16 limbs each:
// X, Y, ZZ, ZZZ) into all[world_size][count][16].  The prover calls it three times per proof (7, 2 and 9 points), not once per
// commitment.  Single-GPU provers pass nothing.
using AllGather = std::function<void(const uint64_t* mine, uint64_t count, uint64_t* all)>;

// KZGProverKey (src/prover.rs:32): owns a plk_ctx with the powers resident in HBM.
class DeviceInputs;
class KZGProverKey {
public:
    explicit KZGProverKey(int32_t device);
    ~KZGProverKey();
    KZGProverKey(const KZGProverKey&) = delete;
    KZGProverKey& operator=(const KZGProverKey&) = delete;
    // PCS::trim output: powers_of_g[0..len), powers_of_gamma_g[0..3)
    void load(const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy);
    // PCS::setup stand-in with a caller-chosen trapdoor (synthetic SRS)
    void generate(const Fr& tau, const Fr& gscal, uint64_t len);
    // this rank keeps powers [first, first + count) of a total of len
    void generate_shard(const Fr& tau, const Fr& gscal, uint64_t first, uint64_t count, uint64_t len);
    // this rank keeps every world_size-th block of 2^log_block powers (block b belongs to rank b mod world_size): unlike contiguous
    // ranges this spreads the prover's n-coefficient polynomials over all ranks although the SRS has 4n + 1 powers
    void generate_cyclic(const Fr& tau, const Fr& gscal, uint32_t rank, uint32_t world_size, uint32_t log_block, uint64_t len);
    void load_cyclic(const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy, uint32_t rank, uint32_t world_size, uint32_t log_block);
    // PCS::setup(max_degree, None, rng) followed by PCS::trim(&pp, max_degree, 1, None) (src/main.rs:182-183): draws beta, g, gamma_g, h
    // from rng as KZG10::setup does, computes the max_degree + 1 powers on the GPU and returns the verifier key
    KZGVerifierKey setup(uint64_t max_degree, StdRng& rng);
    KZGVerifierKey setup_cyclic(uint64_t max_degree, StdRng& rng, uint32_t rank, uint32_t world_size, uint32_t log_block);
    void set_sharding(uint32_t rank, uint32_t world_size, AllGather gather);
    plk_ctx* ctx() const { return ctx_; }
    uint64_t supported_degree() const { return len_ ? len_ - 1 : 0; }
    // the value PCS::open squeezes from its Poseidon sponge (src/prover.rs:285-303): 128 bits, supplied by the caller
    Fr opening_challenge{};
    uint32_t rank = 0, world_size = 1;
    AllGather gather;
    // device buffers that proofs from host arrays stream their inputs into; kept between proofs, grown when a proof needs more
    DeviceInputs& arena(const Domain& domain, const uint64_t len[7]) const;
private:
    plk_ctx* ctx_ = nullptr;
    uint64_t len_ = 0;
    mutable DeviceInputs* arena_ = nullptr;
};

// The prover's inputs resident in HBM (what create_plonk_proof_with_transcript borrows: seven polynomials,
// witness_flat, sigma).  Built once, reused for any number of proofs.
// The prover's arguments as the reference passes them: borrowed host memory (&DensePolynomial, &[Fr], &[usize], src/prover.rs:121-130).
// len[i] counts coefficients after trailing zeros were stripped (DensePolynomial::from_coefficients_vec).
struct HostInputs {
    const Fr* poly[7];      // a, b, c, q_add, q_mul, s_id, s_sigma
    uint64_t len[7];
    const Fr* witness_flat; // 3n
    const uint64_t* sigma;  // 3n
    Domain domain;
};
class DeviceInputs {
public:
    DeviceInputs(const KZGProverKey& pk, const DensePolynomial& a, const DensePolynomial& b, const DensePolynomial& c, const DensePolynomial& q_add,
                 const DensePolynomial& q_mul, const DensePolynomial& s_id, const DensePolynomial& s_sigma, const std::vector<Fr>& witness_flat,
                 const std::vector<uint64_t>& sigma, const Domain& domain);
    // buffers only, for at most cap[i] coefficients each: filled by the prover itself from a HostInputs while it runs
    DeviceInputs(const KZGProverKey& pk, const Domain& domain, const uint64_t cap[7]);
    ~DeviceInputs();
    uint64_t cap[7] = {0, 0, 0, 0, 0, 0, 0};
    const HostInputs* host = nullptr;   // set while a proof streams its inputs in
    DeviceInputs(const DeviceInputs&) = delete;
    Domain domain;
    uint64_t len[7];
    void* poly[7];          // a, b, c, q_add, q_mul, s_id, s_sigma
    void* witness_flat;
    void* sigma;
    void* work[5];          // q_add values, q_mul values, z coefficients, z values, t
private:
    plk_ctx* ctx_;
};

// src/prover.rs:119-133
PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, const DensePolynomial& a, const DensePolynomial& b, const DensePolynomial& c,
                                              const DensePolynomial& q_add, const DensePolynomial& q_mul, const DensePolynomial& s_id,
                                              const DensePolynomial& s_sigma, const std::vector<Fr>& public_inputs, const std::vector<Fr>& witness_flat,
                                              const std::vector<uint64_t>& sigma, const Domain& domain, Rng& rng, ProveTrace* trace = nullptr);
// the same on borrowed host arrays (no copy on the host): the polynomials cross PCIe while the first commitments are computed
PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, const HostInputs& host, const std::vector<Fr>& public_inputs, Rng& rng,
                                              ProveTrace* trace = nullptr);
// the same on inputs that already live in HBM
PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, DeviceInputs& in, const std::vector<Fr>& public_inputs, Rng& rng,
                                              ProveTrace* trace = nullptr);

// src/circuit.rs:491-498 (s_id_vals[j] = j as in src/prover.rs:205)
DensePolynomial build_grand_product(const KZGProverKey& pk, const std::vector<Fr>& witness_flat, const std::vector<uint64_t>& sigma, const Domain& domain,
                                    const Fr& beta, const Fr& gamma);
// src/circuit.rs:442 with the witness table and permutation argument passed explicitly
DensePolynomial build_quotient_polynomial(const KZGProverKey& pk, const std::vector<Fr>& a_col, const std::vector<Fr>& b_col, const std::vector<Fr>& c_col,
                                          const std::vector<Fr>& q_add_vals, const std::vector<Fr>& q_mul_vals, const std::vector<Fr>& z_vals,
                                          const std::vector<uint64_t>& sigma, const Domain& domain, const Fr& beta, const Fr& gamma);
// src/poly_utils.rs:7 and :14
DensePolynomial interpolate_selector(const KZGProverKey& pk, const std::vector<Fr>& values);
std::pair<DensePolynomial, DensePolynomial> interpolate_permutation_polynomials(const KZGProverKey& pk, const std::vector<uint64_t>& sigma, const Domain& domain);

// ---- verifier (src/verifier.rs:57-236); CPU only
// G2 point: x and y in Fq2 as c0 || c1 (each x * 2^256 mod q)
struct G2Affine {
    uint64_t x[8], y[8];
    int32_t infinity;
};
// kzg10::VerifierKey { g, gamma_g, h, beta_h }
struct KZGVerifierKey {
    G1Affine g, gamma_g;
    G2Affine h, beta_h;
};
G2Affine g2_rand(StdRng& rng);   // G2Projective::rand(rng), affine
// verifier key of the synthetic SRS of KZGProverKey::generate (what PCS::setup derives from its rng draws, src/main.rs:182)
KZGVerifierKey verifier_key_from_trapdoor(const Fr& tau, const Fr& gscal);
// src/verifier.rs:57: replays the transcript, then PCS::check for the nine openings (two pairings each).  opening_challenge is the
// value the prover's and verifier's Poseidon sponges squeeze (src/verifier.rs:183-191), supplied by the caller as in the prover.
bool verify_plonk_proof_with_transcript(const KZGVerifierKey& vk, const PlonkProof& proof, const std::vector<Fr>& public_inputs, const Fr& opening_challenge);
// inverse of serialize_proof (ark-serialize deserialize_compressed with validation of every field element and point)
PlonkProof deserialize_proof(const uint8_t* bytes, size_t len);
bool pairing_selftest(const Fr& a, const Fr& b);

// canonical proof bytes: the fields of PlonkProof in declaration order, each with ark-serialize's compressed encoding
// (Fr 32 B; Commitment 33 B; Proof 65 B) = 1170 bytes
std::vector<uint8_t> serialize_proof(const PlonkProof& p);
void serialize_fr(const Fr& x, uint8_t out[32]);
void serialize_g1_compressed(const G1Affine& p, uint8_t out[32]);

// src/transcript.rs:55-157 over merlin::Transcript
class PlonkTranscript {
public:
    explicit PlonkTranscript(const char* domain_separator);
    void append_message(const char* label, const uint8_t* msg, size_t len);
    void challenge_bytes(const char* label, uint8_t* out, size_t len);
    void absorb_field(const char* label, const Fr& x);
    void absorb_commitment(const char* label, const KZGCommitment& c);
    void absorb_public_inputs(const std::vector<Fr>& public_inputs);
    void absorb_evaluations(const Fr& a, const Fr& b, const Fr& c, const Fr& q_add, const Fr& q_mul, const Fr& s_id, const Fr& s_sigma, const Fr& z, const Fr& t);
    Fr challenge_scalar(const char* label);
    Fr challenge_beta() { return challenge_scalar("beta"); }
    Fr challenge_gamma() { return challenge_scalar("gamma"); }
    Fr challenge_alpha() { return challenge_scalar("alpha"); }
    Fr challenge_zeta() { return challenge_scalar("zeta"); }
private:
    void begin_op(uint8_t flags, bool more);
    void absorb(const uint8_t* d, size_t n);
    void squeeze(uint8_t* d, size_t n);
    void run_f();
    void meta_ad(const uint8_t* d, size_t n, bool more);
    uint8_t st_[200];
    uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
};

}  // namespace plonk
