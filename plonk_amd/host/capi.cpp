// C entry points of libplonk_host.so for the Python harness (tests, bench.py, smoke).  They only marshal arguments into
// the C++ surface of plonk_host.hpp; failures come back as a status code plus the reference-style message.
#include "plonk_host.hpp"
#include "circuit.hpp"
#include "../../include/plonk_host.h"   // the public declarations of what is defined here: a mismatch does not compile
#include <string.h>
#include <string>

using namespace plonk;

namespace {
thread_local std::string g_err;
template <class F> int32_t guarded(F&& f) {
    try { f(); g_err.clear(); return PLK_OK; }
    catch (const Error& e) { g_err = e.what(); return e.status ? e.status : PLK_ERR_INVALID_ARG; }
    catch (const std::exception& e) { g_err = e.what(); return PLK_ERR_INVALID_ARG; }
}
DensePolynomial poly_of(const uint64_t* p, uint64_t len) {
    std::vector<Fr> c(len);
    if (len) memcpy(c.data(), p, len * 32);
    return DensePolynomial::from_coefficients_vec(std::move(c));
}
}  // namespace

extern "C" {

const char* plonk_host_last_error() { return g_err.c_str(); }

int32_t plonk_pk_create(int32_t device, void** out) { return guarded([&] { *out = new KZGProverKey(device); }); }
int32_t plonk_pk_destroy(void* pk) { return guarded([&] { delete static_cast<KZGProverKey*>(pk); }); }
void* plonk_pk_ctx(void* pk) { return static_cast<KZGProverKey*>(pk)->ctx(); }
int32_t plonk_pk_load(void* pk, const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy) {
    return guarded([&] { static_cast<KZGProverKey*>(pk)->load(g_xy, len, gamma_g_xy); });
}
int32_t plonk_pk_generate(void* pk, const uint64_t* tau, const uint64_t* gscal, uint64_t first, uint64_t count, uint64_t len) {
    return guarded([&] { Fr t, g; memcpy(t.v, tau, 32); memcpy(g.v, gscal, 32); static_cast<KZGProverKey*>(pk)->generate_shard(t, g, first, count, len); });
}
int32_t plonk_pk_generate_cyclic(void* pk, const uint64_t* tau, const uint64_t* gscal, uint32_t rank, uint32_t world, uint32_t log_block, uint64_t len) {
    return guarded([&] { Fr t, g; memcpy(t.v, tau, 32); memcpy(g.v, gscal, 32); static_cast<KZGProverKey*>(pk)->generate_cyclic(t, g, rank, world, log_block, len); });
}
int32_t plonk_pk_set_opening_challenge(void* pk, const uint64_t* c) { return guarded([&] { memcpy(static_cast<KZGProverKey*>(pk)->opening_challenge.v, c, 32); }); }
// gather(user, mine[count][16], count, all[world][count][16])
typedef void (*plonk_gather_fn)(void* user, const uint64_t* mine, uint64_t count, uint64_t* all);
int32_t plonk_pk_set_sharding(void* pk, uint32_t rank, uint32_t world, plonk_gather_fn fn, void* user) {
    return guarded([&] { static_cast<KZGProverKey*>(pk)->set_sharding(rank, world, [fn, user](const uint64_t* m, uint64_t k, uint64_t* a) { fn(user, m, k, a); }); });
}

// Sharding with the exchange inside the library: RCCL communicator on the key's context (plk_comm_init), all-gather of the partial sums
// through plk_comm_all_gather_partials.  id128: what rank 0 obtained from plonk_comm_get_unique_id and handed to every rank.
int32_t plonk_comm_get_unique_id(uint8_t* id128) { return plk_comm_get_unique_id(id128); }
int32_t plonk_pk_comm_init(void* pk, const uint8_t* id128, uint32_t rank, uint32_t world) {
    return guarded([&] {
        KZGProverKey* k = static_cast<KZGProverKey*>(pk);
        plk_ctx* ctx = k->ctx();
        const int32_t rc = plk_comm_init(ctx, id128, rank, world);
        if (rc != PLK_OK) throw Error(rc, std::string("plk_comm_init: ") + plk_last_error(ctx));
        k->set_sharding(rank, world, [ctx](const uint64_t* mine, uint64_t count, uint64_t* all) {
            const int32_t rc2 = plk_comm_all_gather_partials(ctx, mine, count, all);
            if (rc2 != PLK_OK) throw Error(rc2, std::string("plk_comm_all_gather_partials: ") + plk_last_error(ctx));
        });
    });
}

// polys: a, b, c, q_add, q_mul, s_id, s_sigma (coefficients, any trailing zeros are stripped as from_coefficients_vec does)
int32_t plonk_inputs_create(void* pk, const uint64_t* const* polys, const uint64_t* lens, const uint64_t* witness_flat, const uint64_t* sigma, uint32_t log_n,
                            void** out) {
    return guarded([&] {
        const Domain d{log_n};
        DensePolynomial p[7];
        for (int i = 0; i < 7; ++i) p[i] = poly_of(polys[i], lens[i]);
        std::vector<Fr> w(3 * d.size());
        memcpy(w.data(), witness_flat, w.size() * 32);
        std::vector<uint64_t> s(sigma, sigma + 3 * d.size());
        *out = new DeviceInputs(*static_cast<KZGProverKey*>(pk), p[0], p[1], p[2], p[3], p[4], p[5], p[6], w, s, d);
    });
}
int32_t plonk_inputs_destroy(void* in) { return guarded([&] { delete static_cast<DeviceInputs*>(in); }); }

// blinds: the Fr::rand draws in order (27 for a full proof).  proof_bytes: 1170 bytes.  challenges: beta, gamma, alpha, zeta.  seconds[6].
int32_t plonk_prove(void* pk, void* inputs, const uint64_t* public_inputs, uint64_t n_public, const uint64_t* blinds, uint64_t n_blinds, uint8_t* proof_bytes,
                    uint64_t* challenges, double* seconds) {
    return guarded([&] {
        std::vector<Fr> pub(n_public);
        if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
        uint64_t next = 0;
        Rng rng = [&]() -> Fr {
            if (next >= n_blinds) throw Error(PLK_ERR_INVALID_ARG, "rng exhausted: the prover draws 27 field elements");
            Fr r; memcpy(r.v, blinds + 4 * next, 32); ++next; return r;
        };
        ProveTrace tr;
        const PlonkProof proof = create_plonk_proof_with_transcript(*static_cast<KZGProverKey*>(pk), *static_cast<DeviceInputs*>(inputs), pub, rng, &tr);
        const std::vector<uint8_t> bytes = serialize_proof(proof);
        memcpy(proof_bytes, bytes.data(), bytes.size());
        if (challenges) { memcpy(challenges, tr.beta.v, 32); memcpy(challenges + 4, tr.gamma.v, 32); memcpy(challenges + 8, tr.alpha.v, 32); memcpy(challenges + 12, tr.zeta.v, 32); }
        if (seconds) memcpy(seconds, tr.seconds, sizeof(tr.seconds));
    });
}

// plonk_prove on host arrays (the reference's call shape, src/prover.rs:119-133): polys as in plonk_inputs_create, borrowed, not copied.
int32_t plonk_prove_host(void* pk, const uint64_t* const* polys, const uint64_t* lens, const uint64_t* witness_flat, const uint64_t* sigma, uint32_t log_n,
                         const uint64_t* public_inputs, uint64_t n_public, const uint64_t* blinds, uint64_t n_blinds, uint8_t* proof_bytes, uint64_t* challenges,
                         double* seconds) {
    return guarded([&] {
        HostInputs host;
        for (int i = 0; i < 7; ++i) {
            const Fr* p = reinterpret_cast<const Fr*>(polys[i]);
            uint64_t len = lens[i];
            while (len && p[len - 1].is_zero()) --len;   // from_coefficients_vec
            host.poly[i] = p; host.len[i] = len;
        }
        host.witness_flat = reinterpret_cast<const Fr*>(witness_flat);
        host.sigma = sigma;
        host.domain = Domain{log_n};
        std::vector<Fr> pub(n_public);
        if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
        uint64_t next = 0;
        Rng rng = [&]() -> Fr {
            if (next >= n_blinds) throw Error(PLK_ERR_INVALID_ARG, "rng exhausted: the prover draws 27 field elements");
            Fr r; memcpy(r.v, blinds + 4 * next, 32); ++next; return r;
        };
        ProveTrace tr;
        const PlonkProof proof = create_plonk_proof_with_transcript(*static_cast<KZGProverKey*>(pk), host, pub, rng, &tr);
        const std::vector<uint8_t> bytes = serialize_proof(proof);
        memcpy(proof_bytes, bytes.data(), bytes.size());
        if (challenges) { memcpy(challenges, tr.beta.v, 32); memcpy(challenges + 4, tr.gamma.v, 32); memcpy(challenges + 8, tr.alpha.v, 32); memcpy(challenges + 12, tr.zeta.v, 32); }
        if (seconds) memcpy(seconds, tr.seconds, sizeof(tr.seconds));
    });
}

// src/verifier.rs:57 on serialized proof bytes, with the verifier key of the synthetic SRS (tau, gscal).  *ok = 1 when all nine openings check.
int32_t plonk_verify(const uint64_t* tau, const uint64_t* gscal, const uint8_t* proof_bytes, uint64_t proof_len, const uint64_t* public_inputs, uint64_t n_public,
                     const uint64_t* opening_challenge, int32_t* ok) {
    return guarded([&] {
        Fr t, g, c;
        memcpy(t.v, tau, 32); memcpy(g.v, gscal, 32); memcpy(c.v, opening_challenge, 32);
        std::vector<Fr> pub(n_public);
        if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
        const PlonkProof proof = deserialize_proof(proof_bytes, proof_len);
        *ok = verify_plonk_proof_with_transcript(verifier_key_from_trapdoor(t, g), proof, pub, c) ? 1 : 0;
    });
}
// ---- the reference's rng, setup and opening challenge (src/main.rs:25,182-183; src/prover.rs:36-59)
int32_t plonk_rng_create(const uint8_t* seed32, void** out) { return guarded([&] { *out = seed32 ? new StdRng(seed32) : new StdRng(test_rng()); }); }
int32_t plonk_rng_destroy(void* rng) { return guarded([&] { delete static_cast<StdRng*>(rng); }); }
uint64_t plonk_rng_next_u64(void* rng) { return static_cast<StdRng*>(rng)->next_u64(); }
uint32_t plonk_rng_next_u32(void* rng) { return static_cast<StdRng*>(rng)->next_u32(); }
void plonk_rng_fr(void* rng, uint64_t n, uint64_t* out) {
    for (uint64_t i = 0; i < n; ++i) { const Fr x = fr_rand(*static_cast<StdRng*>(rng)); memcpy(out + 4 * i, x.v, 32); }
}
void plonk_rng_g1(void* rng, uint64_t* xy) { const G1Affine p = g1_rand(*static_cast<StdRng*>(rng)); memcpy(xy, p.xy, 64); }
void plonk_rng_g2(void* rng, uint64_t* x8y8) { const G2Affine p = g2_rand(*static_cast<StdRng*>(rng)); memcpy(x8y8, p.x, 64); memcpy(x8y8 + 8, p.y, 64); }
void plonk_chacha_block(const uint8_t* key32, uint64_t counter, uint64_t stream, int32_t rounds, uint32_t* out16) { chacha_block(key32, counter, stream, rounds, out16); }
void plonk_test_sponge_challenge(uint64_t* out4) { const Fr c = test_sponge_challenge(); memcpy(out4, c.v, 32); }
// vk48: g (8), gamma_g (8), h (x 8, y 8), beta_h (x 8, y 8)
static void vk_store(const KZGVerifierKey& vk, uint64_t* o) {
    memcpy(o, vk.g.xy, 64); memcpy(o + 8, vk.gamma_g.xy, 64); memcpy(o + 16, vk.h.x, 64); memcpy(o + 24, vk.h.y, 64); memcpy(o + 32, vk.beta_h.x, 64); memcpy(o + 40, vk.beta_h.y, 64);
}
static KZGVerifierKey vk_load(const uint64_t* o) {
    KZGVerifierKey vk{};
    memcpy(vk.g.xy, o, 64); memcpy(vk.gamma_g.xy, o + 8, 64); memcpy(vk.h.x, o + 16, 64); memcpy(vk.h.y, o + 24, 64); memcpy(vk.beta_h.x, o + 32, 64); memcpy(vk.beta_h.y, o + 40, 64);
    return vk;
}
int32_t plonk_pk_setup(void* pk, uint64_t max_degree, void* rng, uint32_t rank, uint32_t world, uint32_t log_block, uint64_t* vk48) {
    return guarded([&] { vk_store(static_cast<KZGProverKey*>(pk)->setup_cyclic(max_degree, *static_cast<StdRng*>(rng), rank, world, log_block), vk48); });
}
// plonk_prove drawing its blinding coefficients from the generator, as the reference's `rng: &mut R` does
int32_t plonk_prove_rng(void* pk, void* inputs, const uint64_t* public_inputs, uint64_t n_public, void* rng, uint8_t* proof_bytes, uint64_t* challenges, double* seconds) {
    return guarded([&] {
        std::vector<Fr> pub(n_public);
        if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
        Rng draw = as_rng(*static_cast<StdRng*>(rng));
        ProveTrace tr;
        const PlonkProof proof = create_plonk_proof_with_transcript(*static_cast<KZGProverKey*>(pk), *static_cast<DeviceInputs*>(inputs), pub, draw, &tr);
        const std::vector<uint8_t> bytes = serialize_proof(proof);
        memcpy(proof_bytes, bytes.data(), bytes.size());
        if (challenges) { memcpy(challenges, tr.beta.v, 32); memcpy(challenges + 4, tr.gamma.v, 32); memcpy(challenges + 8, tr.alpha.v, 32); memcpy(challenges + 12, tr.zeta.v, 32); }
        if (seconds) memcpy(seconds, tr.seconds, sizeof(tr.seconds));
    });
}
// src/verifier.rs:57 with an explicit verifier key (vk48 as plonk_pk_setup returns it)
int32_t plonk_verify_vk(const uint64_t* vk48, const uint8_t* proof_bytes, uint64_t proof_len, const uint64_t* public_inputs, uint64_t n_public,
                        const uint64_t* opening_challenge, int32_t* ok) {
    return guarded([&] {
        Fr c;
        memcpy(c.v, opening_challenge, 32);
        std::vector<Fr> pub(n_public);
        if (n_public) memcpy(pub.data(), public_inputs, n_public * 32);
        *ok = verify_plonk_proof_with_transcript(vk_load(vk48), deserialize_proof(proof_bytes, proof_len), pub, c) ? 1 : 0;
    });
}
// ---- circuit construction (src/circuit.rs) for the tests: rebuilds a circuit through CircuitBuilder from its variable values and
// gates (type 0 Add / 1 Mul, a, b, out per gate; a variable that is no gate's output is a new_variable) and returns the tables of
// Circuit::from_builder, WitnessTable::flatten and compute_sigma_mapping for the domain of 2^log_n rows.
int32_t plonk_circuit_tables(const uint64_t* var_values, uint64_t n_vars, const int64_t* gates, uint64_t n_gates, uint32_t log_n, uint64_t* a_col, uint64_t* b_col,
                             uint64_t* c_col, uint64_t* q_add, uint64_t* q_mul, uint64_t* witness_flat, uint64_t* sigma) {
    return guarded([&] {
        CircuitBuilder b;
        uint64_t next_gate = 0;
        for (uint64_t v = 0; v < n_vars; ++v) {
            if (next_gate < n_gates && (uint64_t)gates[4 * next_gate + 3] == v) {
                const int64_t* g = gates + 4 * next_gate++;
                const Variable out = b.add_gate(g[0] == 0 ? GateType::Add : GateType::Mul, Variable{(size_t)g[1]}, Variable{(size_t)g[2]});
                if (out.index != v || memcmp(b.variables[v].v, var_values + 4 * v, 32) != 0) throw Error(PLK_ERR_INVALID_ARG, "gate output does not match the given value");
            } else {
                Fr x; memcpy(x.v, var_values + 4 * v, 32);
                b.new_variable(&x);
            }
        }
        if (next_gate != n_gates) throw Error(PLK_ERR_INVALID_ARG, "gates are not in the order of their outputs");
        const Domain domain{log_n};
        const Circuit c = Circuit::from_builder(std::move(b), domain);
        const size_t n = domain.size();
        if (c.witness.a_col.size() != n) throw Error(PLK_ERR_INVALID_ARG, "more gates than rows");
        memcpy(a_col, c.witness.a_col.data(), n * 32); memcpy(b_col, c.witness.b_col.data(), n * 32); memcpy(c_col, c.witness.c_col.data(), n * 32);
        memcpy(q_add, c.witness.q_add.data(), n * 32); memcpy(q_mul, c.witness.q_mul.data(), n * 32);
        const std::vector<Fr> flat = c.witness.flatten();
        memcpy(witness_flat, flat.data(), flat.size() * 32);
        const std::vector<uint64_t> sg = c.permutation.compute_sigma_mapping(n);
        memcpy(sigma, sg.data(), sg.size() * 8);
    });
}
int32_t plonk_pairing_selftest(const uint64_t* a, const uint64_t* b) {
    Fr x, y;
    memcpy(x.v, a, 32); memcpy(y.v, b, 32);
    return pairing_selftest(x, y) ? 1 : 0;
}
// serialize(deserialize(bytes)) for the tests: 0 when the bytes decode and re-encode to themselves
int32_t plonk_proof_roundtrip(const uint8_t* proof_bytes, uint64_t len) {
    return guarded([&] {
        const std::vector<uint8_t> again = serialize_proof(deserialize_proof(proof_bytes, len));
        if (again.size() != len || memcmp(again.data(), proof_bytes, len) != 0) throw Error(PLK_ERR_INVALID_ARG, "proof bytes do not round-trip");
    });
}

void plonk_transcript_kat(uint8_t* out32) {
    PlonkTranscript t("test protocol");
    t.append_message("some label", reinterpret_cast<const uint8_t*>("some data"), 9);
    t.challenge_bytes("challenge", out32, 32);
}
void plonk_transcript_empty_beta(uint64_t* out) { PlonkTranscript t("plonk_proof"); const Fr b = t.challenge_beta(); memcpy(out, b.v, 32); }
void plonk_serialize_g1(const uint64_t* xy, int32_t inf, uint8_t* out32) { G1Affine p; memcpy(p.xy, xy, 64); p.infinity = inf; serialize_g1_compressed(p, out32); }

}  // extern "C"
