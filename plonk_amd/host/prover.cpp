// Round-by-round prover of the reference (src/prover.rs:119-336) above libplonk_hip: the transcript, the rng draws and the
// serialization run here on the host, every polynomial stays in HBM and only 32..128-byte results cross PCIe.
#include "plonk_host.hpp"
#include <string.h>
#include <chrono>
#include <algorithm>

namespace plonk {

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void check(plk_ctx* ctx, int32_t rc) {
    if (rc == PLK_OK) return;
    std::string detail = plk_last_error(ctx);
    switch (rc) {
        case PLK_ERR_NOT_DIVISIBLE: throw Error(rc, "t(X) not divisible by Z_H(X)");                           // src/circuit.rs:487
        case PLK_ERR_INVERSE_OF_ZERO: throw Error(rc, "called `Option::unwrap()` on a `None` value: " + detail);  // .inverse().unwrap()
        case PLK_ERR_INDEX_OUT_OF_RANGE: throw Error(rc, "index out of bounds: " + detail);
        case PLK_ERR_SRS_TOO_SHORT: throw Error(rc, "called `Result::unwrap()` on an `Err` value: TooManyCoefficients: " + detail);
        default: throw Error(rc, "libplonk_hip: " + detail);
    }
}

struct DevBuf {
    plk_ctx* ctx;
    void* p = nullptr;
    DevBuf(plk_ctx* c, uint64_t bytes) : ctx(c) { check(c, plk_malloc(c, bytes, &p)); }
    ~DevBuf() { if (p) plk_free(ctx, p); }
    DevBuf(const DevBuf&) = delete;
};


}  // namespace

KZGProverKey::KZGProverKey(int32_t device) {
    const int32_t rc = plk_create(device, &ctx_);
    if (rc != PLK_OK) throw Error(rc, "plk_create failed: no usable GPU (there is no CPU fallback)");
}
KZGProverKey::~KZGProverKey() {
    delete arena_;
    if (ctx_) plk_destroy(ctx_);
}
DeviceInputs& KZGProverKey::arena(const Domain& domain, const uint64_t len[7]) const {
    bool fits = arena_ && arena_->domain.log_size == domain.log_size;
    for (int i = 0; i < 7 && fits; ++i) fits = len[i] <= arena_->cap[i];
    if (!fits) {
        uint64_t cap[7];
        for (int i = 0; i < 7; ++i) cap[i] = std::max<uint64_t>(len[i], arena_ && arena_->domain.log_size == domain.log_size ? arena_->cap[i] : 0);
        delete arena_;
        arena_ = nullptr;
        arena_ = new DeviceInputs(*this, domain, cap);
    }
    return *arena_;
}
void KZGProverKey::load(const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy) {
    check(ctx_, plk_srs_load(ctx_, g_xy, len, gamma_g_xy));
    len_ = len;
}
void KZGProverKey::generate(const Fr& tau, const Fr& gscal, uint64_t len) {
    check(ctx_, plk_srs_generate(ctx_, tau.v, gscal.v, len));
    len_ = len;
}
void KZGProverKey::generate_shard(const Fr& tau, const Fr& gscal, uint64_t first, uint64_t count, uint64_t len) {
    check(ctx_, plk_srs_generate_shard(ctx_, tau.v, gscal.v, first, count, len));
    len_ = len;
}
void KZGProverKey::generate_cyclic(const Fr& tau, const Fr& gscal, uint32_t r, uint32_t w, uint32_t log_block, uint64_t len) {
    check(ctx_, plk_srs_generate_cyclic(ctx_, tau.v, gscal.v, r, w, log_block, len));
    len_ = len;
}
void KZGProverKey::load_cyclic(const uint64_t* g_xy, uint64_t len, const uint64_t* gamma_g_xy, uint32_t r, uint32_t w, uint32_t log_block) {
    check(ctx_, plk_srs_load_cyclic(ctx_, g_xy, len, gamma_g_xy, r, w, log_block));
    len_ = len;
}
void KZGProverKey::set_sharding(uint32_t r, uint32_t w, AllGather g) { rank = r; world_size = w; gather = std::move(g); }

DeviceInputs::DeviceInputs(const KZGProverKey& pk, const DensePolynomial& a, const DensePolynomial& b, const DensePolynomial& c, const DensePolynomial& q_add,
                           const DensePolynomial& q_mul, const DensePolynomial& s_id, const DensePolynomial& s_sigma, const std::vector<Fr>& wflat,
                           const std::vector<uint64_t>& sg, const Domain& dom)
    : domain(dom), ctx_(pk.ctx()) {
    const uint64_t n = dom.size();
    if (wflat.size() != 3 * n || sg.size() != 3 * n) throw Error(PLK_ERR_INVALID_ARG, "witness_flat and sigma must have 3 * domain.size() entries");
    const DensePolynomial* polys[7] = {&a, &b, &c, &q_add, &q_mul, &s_id, &s_sigma};
    for (int i = 0; i < 7; ++i) { poly[i] = nullptr; len[i] = polys[i]->coeffs.size(); }
    witness_flat = sigma = nullptr;
    for (auto& w : work) w = nullptr;
    for (int i = 0; i < 7; ++i) {
        // selectors are padded to n so that they can be transformed in place copies later
        const uint64_t cap = (i == 3 || i == 4) ? std::max<uint64_t>(len[i], n) : std::max<uint64_t>(len[i], 1);
        check(ctx_, plk_malloc(ctx_, cap * 32, &poly[i]));
        if (len[i]) check(ctx_, plk_upload(ctx_, poly[i], polys[i]->coeffs.data(), len[i] * 32));
    }
    check(ctx_, plk_malloc(ctx_, 3 * n * 32, &witness_flat));
    check(ctx_, plk_upload(ctx_, witness_flat, wflat.data(), 3 * n * 32));
    check(ctx_, plk_malloc(ctx_, 3 * n * 8, &sigma));
    check(ctx_, plk_upload(ctx_, sigma, sg.data(), 3 * n * 8));
    for (auto& w : work) check(ctx_, plk_malloc(ctx_, n * 32, &w));
}
DeviceInputs::DeviceInputs(const KZGProverKey& pk, const Domain& dom, const uint64_t caps[7]) : domain(dom), ctx_(pk.ctx()) {
    const uint64_t n = dom.size();
    for (int i = 0; i < 7; ++i) { poly[i] = nullptr; len[i] = 0; }
    witness_flat = sigma = nullptr;
    for (auto& w : work) w = nullptr;
    for (int i = 0; i < 7; ++i) {
        cap[i] = (i == 3 || i == 4) ? std::max<uint64_t>(caps[i], n) : std::max<uint64_t>(caps[i], 1);
        check(ctx_, plk_malloc(ctx_, cap[i] * 32, &poly[i]));
    }
    check(ctx_, plk_malloc(ctx_, 3 * n * 32, &witness_flat));
    check(ctx_, plk_malloc(ctx_, 3 * n * 8, &sigma));
    for (auto& w : work) check(ctx_, plk_malloc(ctx_, n * 32, &w));
}
DeviceInputs::~DeviceInputs() {
    for (auto p : poly) if (p) plk_free(ctx_, p);
    if (witness_flat) plk_free(ctx_, witness_flat);
    if (sigma) plk_free(ctx_, sigma);
    for (auto w : work) if (w) plk_free(ctx_, w);
}

namespace {

// Ends `count` tickets and returns their points.  With a sharded SRS every ticket yields this rank's partial sum; all partial sums
// of the call travel in one collective and are added on the host (EC addition is not an RCCL reduction).
// `order`: the sequence in which the tickets were begun (nullptr: as listed).  Every end finishes its MSM on the CPU (about a hundred
// point additions), so the tickets whose results have arrived are ended first, while the GPU still works on the others; when none
// is ready the one begun earliest is waited for.
void finish_points(const KZGProverKey& pk, const int32_t* tickets, int count, bool is_open, G1Affine* out, Fr* random_v, Fr* eval, const int* order = nullptr) {
    plk_ctx* ctx = pk.ctx();
    // the tickets in the sequence in which they were begun: when nothing has arrived yet, the library waits for the first one listed
    std::vector<int32_t> t(count);
    std::vector<int> who(count);
    for (int k = 0; k < count; ++k) { who[k] = order ? order[k] : k; t[k] = tickets[who[k]]; }
    std::vector<uint64_t> pts((size_t)16 * count), rv((size_t)4 * count), ev((size_t)4 * count);
    std::vector<int32_t> inf(count);
    const bool sharded = pk.world_size > 1;
    check(ctx, plk_kzg_end_many(ctx, (uint32_t)count, t.data(), sharded ? 0 : 1, pts.data(), inf.data(), rv.data(), ev.data()));
    if (is_open)
        for (int k = 0; k < count; ++k) { memcpy(random_v[who[k]].v, &rv[(size_t)4 * k], 32); memcpy(eval[who[k]].v, &ev[(size_t)4 * k], 32); }
    if (!sharded) {
        for (int k = 0; k < count; ++k) { memcpy(out[who[k]].xy, &pts[(size_t)16 * k], 64); out[who[k]].infinity = inf[k]; }
        return;
    }
    std::vector<uint64_t> all((size_t)16 * count * pk.world_size), xy((size_t)8 * count);
    pk.gather(pts.data(), (uint64_t)count, all.data());
    check(ctx, plk_g1_sum_partials_many(all.data(), pk.world_size, (uint64_t)count, xy.data(), inf.data()));
    for (int k = 0; k < count; ++k) { memcpy(out[who[k]].xy, &xy[(size_t)8 * k], 64); out[who[k]].infinity = inf[k]; }
}
// Order in which the independent MSMs of one round are enqueued (results do not depend on it).  One ordinary-sized MSM leads so
// that the accumulation stream starts after a short sort; the MSMs of a handful of terms follow at once, so that their
// reductions are long done when the round ends instead of queueing behind everybody else's on the reduction stream; the rest
// longest first: the 4n-sized s_sigma is then in the middle of the round (its sort runs on one of the two front streams while the
// other feeds the accumulation), and the round ends with an ordinary MSM, whose reduction is the short one.
void msm_order(int* order, int count, const uint64_t* len) {
    std::vector<int> tiny, rest;
    for (int i = 1; i < count; ++i) (len[i] <= 16 ? tiny : rest).push_back(i);
    std::stable_sort(rest.begin(), rest.end(), [&](int x, int y) { return len[x] > len[y]; });
    int k = 0;
    order[k++] = 0;
    for (int i : tiny) order[k++] = i;
    for (int i : rest) order[k++] = i;
}
G1Affine finish_point(const KZGProverKey& pk, int32_t ticket) {
    G1Affine out;
    finish_points(pk, &ticket, 1, false, &out, nullptr, nullptr);
    return out;
}

}  // namespace

static PlonkProof prove_rounds(const KZGProverKey& pk, DeviceInputs& in, const std::vector<Fr>& public_inputs, Rng& rng, ProveTrace* trace);

// The reference panics out of a failed proof and everything it had begun is dropped.  Here a failure is an exception the caller may
// survive: the commitments and openings already enqueued on the GPU are given up first, so that the key can prove again.
PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, DeviceInputs& in, const std::vector<Fr>& public_inputs, Rng& rng, ProveTrace* trace) {
    try {
        return prove_rounds(pk, in, public_inputs, rng, trace);
    } catch (...) {
        plk_kzg_cancel(pk.ctx());
        throw;
    }
}

static PlonkProof prove_rounds(const KZGProverKey& pk, DeviceInputs& in, const std::vector<Fr>& public_inputs, Rng& rng, ProveTrace* trace) {
    plk_ctx* ctx = pk.ctx();
    const uint64_t n = in.domain.size();
    const uint32_t log_n = in.domain.log_size;
    PlonkProof proof;
    memset(&proof, 0, sizeof(proof));
    KZGCommitment* comm = &proof.comm_a;       // a, b, c, q_add, q_mul, s_id, s_sigma, z, t
    KZGOpeningProof* open = &proof.proof_a;
    Fr* eval = &proof.eval_a;
    Fr blind[9][3];                             // kzg10::Randomness of each commitment (3 coefficients: hiding bound 1)
    double t0 = now_s();
    const double t_start = t0;
    auto lap = [&](int i) { const double t1 = now_s(); if (trace) trace->seconds[i] = t1 - t0; t0 = t1; };

    PlonkTranscript transcript("plonk_proof");                                  // src/prover.rs:135
    transcript.absorb_public_inputs(public_inputs);                             // :138

    // ---- round 1: commit a, b, c, q_add, q_mul, s_id, s_sigma :150-156.  The rng is drawn in that order (each commit draws its
    // three blinding coefficients); the seven independent MSMs are then enqueued longest first and collected afterwards.
    {
        for (int i = 0; i < 7; ++i)
            for (auto& r : blind[i]) r = rng();
        int order[7] = {0, 1, 2, 3, 4, 5, 6};
        msm_order(order, 7, in.len);
        if (in.host) {
            // Inputs still on the host: they cross PCIe in the order in which they are committed, shortest first, so that the first
            // sort starts after one polynomial's transfer and the accumulation kernels of the short ones cover the transfer of the
            // 4n-coefficient s_sigma (plk_upload_enqueue orders each sort behind its own polynomial only).
            std::vector<int> tiny, rest;
            for (int i = 0; i < 7; ++i) (in.len[i] <= 16 ? tiny : rest).push_back(i);
            std::stable_sort(rest.begin(), rest.end(), [&](int x, int y) { return in.len[x] < in.len[y]; });
            int k = 0;   // an ordinary polynomial leads, the tiny ones follow it (msm_order's reasons)
            if (!rest.empty()) order[k++] = rest[0];
            for (int i : tiny) order[k++] = i;
            for (size_t r = 1; r < rest.size(); ++r) order[k++] = rest[r];
        }
        int32_t ticket[7];
        if (in.host) {
            for (int k = 0; k < 7; ++k) {
                const int i = order[k];
                check(ctx, plk_upload_enqueue(ctx, in.poly[i], in.host->poly[i], in.len[i] * 32));
                check(ctx, plk_kzg_commit_begin(ctx, in.poly[i], in.len[i], &blind[i][0].v[0], &ticket[i]));
            }
        } else {
            // one call for the round: the library sees all seven lengths and places the long s_sigma's sort where it holds up nobody
            const void* p7[7];
            uint64_t l7[7], b7[7][12];
            int32_t in_order[7];
            for (int k = 0; k < 7; ++k) {
                const int i = order[k];
                p7[k] = in.poly[i]; l7[k] = in.len[i];
                memcpy(b7[k], &blind[i][0].v[0], 96);
            }
            check(ctx, plk_kzg_commit_begin_batch(ctx, 7, p7, l7, &b7[0][0], in_order));
            for (int k = 0; k < 7; ++k) ticket[order[k]] = in_order[k];
        }
        if (in.host) {   // needed from round 2 on: behind the polynomials
            check(ctx, plk_upload_enqueue(ctx, in.witness_flat, in.host->witness_flat, 3 * n * 32));
            check(ctx, plk_upload_enqueue(ctx, in.sigma, in.host->sigma, 3 * n * 8));
        }
        check(ctx, plk_kzg_submit(ctx));
        // Selector values on H for round 3 (domain.fft(&q_add.coeffs), :232-233) depend on no challenge.  They are enqueued here on the
        // detached compute stream, which also carries the reductions just submitted: they start when the last reduction of the round
        // has finished and run while the host finishes the commitments and the transcript, and beside the grand product, whose kernels
        // leave most of the chip idle.  (Enqueued behind the sorts instead, they ran on the critical path: measured when the
        // transform kernel was still allocated 176 registers per wave, 16 more than two accumulation waves leave free on a SIMD, so
        // that its workgroups found no room until the round's last accumulation kernel drained.)
        if (in.len[0] >= n && in.len[1] >= n && in.len[2] >= n) {
            check(ctx, plk_compute_detach(ctx, 1));
            check(ctx, plk_copy_dev(ctx, in.work[0], in.poly[3], in.len[3] * 32, n * 32));
            check(ctx, plk_copy_dev(ctx, in.work[1], in.poly[4], in.len[4] * 32, n * 32));
            check(ctx, plk_ntt_dev_enqueue(ctx, in.work[0], log_n, 0, nullptr));            // q_add values :232
            check(ctx, plk_ntt_dev_enqueue(ctx, in.work[1], log_n, 0, nullptr));            // q_mul values :233
            check(ctx, plk_compute_detach(ctx, 2));   // the quotient, their only reader, is enqueued behind them on the detached stream
        }
        G1Affine c7[7];
        finish_points(pk, ticket, 7, false, c7, nullptr, nullptr, order);
        for (int i = 0; i < 7; ++i) comm[i].comm = c7[i];
    }
    transcript.absorb_commitment("q_add", proof.comm_q_add);                    // :168-194
    transcript.absorb_commitment("q_mul", proof.comm_q_mul);
    transcript.absorb_commitment("a", proof.comm_a);
    transcript.absorb_commitment("b", proof.comm_b);
    transcript.absorb_commitment("c", proof.comm_c);
    transcript.absorb_commitment("s_id", proof.comm_s_id);
    transcript.absorb_commitment("s_sigma", proof.comm_s_sigma);
    const Fr beta = transcript.challenge_beta();                                // :197-198
    const Fr gamma = transcript.challenge_gamma();
    lap(0);

    // ---- round 2: grand product z, commit :205-216.  The commitment's MSM runs on its own stream while round 3's
    // transforms and pointwise pass (which need z but not its commitment) use the compute stream.
    // z_vals = domain.fft(&z_poly.coeffs) (:237) are the values the interpolation started from: kept in work[3], no transform.
    void* z_coeffs = in.work[2];
    // z on H (values only; the call returns when they are enqueued, the one inversion of the batch is done on the host inside it).
    check(ctx, plk_grand_product_dev_enqueue(ctx, in.witness_flat, in.sigma, log_n, beta.v, gamma.v, in.work[3], nullptr));
    if (in.len[0] < n || in.len[1] < n || in.len[2] < n)    // :229-231 -> src/circuit.rs:356-358 index past the end of a trimmed column
        throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: a wire polynomial has fewer than domain.size() coefficients");
    // Round 3's kernels (:224-259) read the values, not the coefficients: they go to the detached stream now and run beside the
    // interpolation and the sort of z, and are through before z's accumulation kernel starts (beside it every kernel is slower).
    check(ctx, plk_compute_detach(ctx, 1));                                       // the detached stream waits for the values ...
    check(ctx, plk_compute_detach(ctx, 2));
    check(ctx, plk_copy_dev(ctx, z_coeffs, in.work[3], n * 32, n * 32));
    check(ctx, plk_ntt_dev_enqueue(ctx, z_coeffs, log_n, 1, nullptr));           // z_poly = ifft(z) :208
    check(ctx, plk_compute_detach(ctx, 3));                                       // ... but not for the interpolation
    check(ctx, plk_quotient_dev_enqueue(ctx, in.poly[0], in.poly[1], in.poly[2], in.work[0], in.work[1], in.work[3], in.sigma, log_n, beta.v, gamma.v, in.work[4]));
    // the commitments below are begun "detached": their sorts stay on the stream that carries z's coefficients
    for (auto& r : blind[7]) r = rng();
    // The quotient of this constraint system has degree < n before the division by Z_H, so it is either the zero polynomial or
    // the prover fails (:487).  Its commitment (three blinding terms, drawn here in the reference's rng order) is therefore
    // started now, beside z's, instead of after the divisibility check; a non-zero t would simply be committed again below.
    for (auto& r : blind[8]) r = rng();
    // (z and t begun as one batch, sharing their launches when z is short enough for the narrow plan: 0.04 ms slower at 2^16 gates)
    int32_t z_ticket, t_ticket = -1;
    check(ctx, plk_kzg_commit_begin(ctx, z_coeffs, n, &blind[7][0].v[0], &z_ticket));
    check(ctx, plk_kzg_commit_begin(ctx, in.work[4], 0, &blind[8][0].v[0], &t_ticket));
    lap(1);

    // ---- round 3: quotient t :224-259, on the detached compute stream.  The reductions of z and t (plk_kzg_submit) share that
    // stream: enqueued behind the quotient's kernels, they wait for z's accumulation without holding anything up.
    uint64_t t_len = 0;
    // a failure below leaves both tickets open: the caller of prove_rounds gives them up
    check(ctx, plk_kzg_submit(ctx));
    const int32_t rc_quotient = plk_quotient_dev_finish(ctx, &t_len);
    check(ctx, plk_compute_detach(ctx, 0));
    check(ctx, rc_quotient);
    if (t_len != 0) {
        finish_point(pk, t_ticket);
        check(ctx, plk_kzg_commit_begin(ctx, in.work[4], t_len, &blind[8][0].v[0], &t_ticket));
    }
    {
        // both commitments in one call: with a sharded SRS their partial sums travel in one collective
        const int32_t zt[2] = {z_ticket, t_ticket};
        G1Affine c2[2];
        finish_points(pk, zt, 2, false, c2, nullptr, nullptr);
        comm[7].comm = c2[0];
        comm[8].comm = c2[1];
    }
    transcript.absorb_commitment("z", proof.comm_z);                           // :214-216
    const Fr alpha = transcript.challenge_alpha();                              // :219 (enters only the public-input term, which is empty)
    transcript.absorb_commitment("t", proof.comm_t);                           // :257-259
    const Fr zeta = transcript.challenge_zeta();                                // :262
    lap(2);

    // ---- rounds 4 + 5: evaluations and openings at zeta :266-303.  The division by (X - zeta) yields f(zeta) as its
    // remainder, so each opening also delivers its evaluation; the transcript absorbs them in the reference's order.
    {
        const void* polys[9] = {in.poly[0], in.poly[1], in.poly[2], in.poly[3], in.poly[4], in.poly[5], in.poly[6], z_coeffs, in.work[4]};
        const uint64_t lens[9] = {in.len[0], in.len[1], in.len[2], in.len[3], in.len[4], in.len[5], in.len[6], n, t_len};
        int order[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
        msm_order(order, 9, lens);
        int32_t ticket[9], in_order[9];
        const void* p9[9];
        uint64_t l9[9], b9[9][12];
        for (int k = 0; k < 9; ++k) {
            const int i = order[k];
            p9[k] = polys[i]; l9[k] = lens[i];
            memcpy(b9[k], &blind[i][0].v[0], 96);
        }
        // one call for the round: the nine divisions by (X - zeta) run as one batch of launches, the sorts and accumulations follow in `order`
        check(ctx, plk_kzg_open_begin_batch(ctx, 9, p9, l9, &b9[0][0], zeta.v, pk.opening_challenge.v, in_order));
        for (int k = 0; k < 9; ++k) ticket[order[k]] = in_order[k];
        check(ctx, plk_kzg_submit(ctx));
        G1Affine w9[9];
        Fr rv9[9];
        finish_points(pk, ticket, 9, true, w9, rv9, eval, order);
        for (int i = 0; i < 9; ++i) { open[i].w = w9[i]; open[i].random_v = rv9[i]; }
    }
    transcript.absorb_evaluations(proof.eval_a, proof.eval_b, proof.eval_c, proof.eval_q_add, proof.eval_q_mul, proof.eval_s_id, proof.eval_s_sigma,
                                  proof.eval_z, proof.eval_t);                  // :277-282
    lap(3);
    if (trace) {
        trace->beta = beta; trace->gamma = gamma; trace->alpha = alpha; trace->zeta = zeta;
        trace->seconds[4] = 0;
        trace->seconds[5] = now_s() - t_start;
    }
    return proof;
}

PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, const HostInputs& host, const std::vector<Fr>& public_inputs, Rng& rng, ProveTrace* trace) {
    DeviceInputs& in = pk.arena(host.domain, host.len);
    for (int i = 0; i < 7; ++i) in.len[i] = host.len[i];
    in.host = &host;
    struct Reset { DeviceInputs& d; ~Reset() { d.host = nullptr; } } reset{in};
    return create_plonk_proof_with_transcript(pk, in, public_inputs, rng, trace);
}
PlonkProof create_plonk_proof_with_transcript(const KZGProverKey& pk, const DensePolynomial& a, const DensePolynomial& b, const DensePolynomial& c,
                                              const DensePolynomial& q_add, const DensePolynomial& q_mul, const DensePolynomial& s_id,
                                              const DensePolynomial& s_sigma, const std::vector<Fr>& public_inputs, const std::vector<Fr>& witness_flat,
                                              const std::vector<uint64_t>& sigma, const Domain& domain, Rng& rng, ProveTrace* trace) {
    const uint64_t n = domain.size();
    if (witness_flat.size() != 3 * n || sigma.size() != 3 * n) throw Error(PLK_ERR_INVALID_ARG, "witness_flat and sigma must have 3 * domain.size() entries");
    const DensePolynomial* polys[7] = {&a, &b, &c, &q_add, &q_mul, &s_id, &s_sigma};
    HostInputs host;
    for (int i = 0; i < 7; ++i) {
        uint64_t len = polys[i]->coeffs.size();
        while (len && polys[i]->coeffs[len - 1].is_zero()) --len;   // from_coefficients_vec
        host.poly[i] = polys[i]->coeffs.data();
        host.len[i] = len;
    }
    host.witness_flat = witness_flat.data();
    host.sigma = sigma.data();
    host.domain = domain;
    return create_plonk_proof_with_transcript(pk, host, public_inputs, rng, trace);
}

DensePolynomial build_grand_product(const KZGProverKey& pk, const std::vector<Fr>& witness_flat, const std::vector<uint64_t>& sigma, const Domain& domain,
                                    const Fr& beta, const Fr& gamma) {
    const uint64_t n = domain.size();
    if (witness_flat.size() < 3 * n || sigma.size() < 3 * n) throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: witness_flat / sigma shorter than 3n");
    std::vector<Fr> z(n);
    check(pk.ctx(), plk_grand_product(pk.ctx(), &witness_flat[0].v[0], sigma.data(), domain.log_size, beta.v, gamma.v, &z[0].v[0]));
    return DensePolynomial::from_coefficients_vec(std::move(z));
}
DensePolynomial build_quotient_polynomial(const KZGProverKey& pk, const std::vector<Fr>& a_col, const std::vector<Fr>& b_col, const std::vector<Fr>& c_col,
                                          const std::vector<Fr>& q_add_vals, const std::vector<Fr>& q_mul_vals, const std::vector<Fr>& z_vals,
                                          const std::vector<uint64_t>& sigma, const Domain& domain, const Fr& beta, const Fr& gamma) {
    const uint64_t n = domain.size();
    for (const auto* v : {&a_col, &b_col, &c_col, &q_add_vals, &q_mul_vals, &z_vals})
        if (v->size() < n) throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: column shorter than the domain");
    if (sigma.size() < 3 * n) throw Error(PLK_ERR_INDEX_OUT_OF_RANGE, "index out of bounds: sigma shorter than 3n");
    std::vector<Fr> t(n);
    uint64_t t_len = 0;
    check(pk.ctx(), plk_quotient(pk.ctx(), &a_col[0].v[0], &b_col[0].v[0], &c_col[0].v[0], &q_add_vals[0].v[0], &q_mul_vals[0].v[0], &z_vals[0].v[0], sigma.data(),
                                 domain.log_size, beta.v, gamma.v, &t[0].v[0], &t_len));
    t.resize(t_len);
    return DensePolynomial::from_coefficients_vec(std::move(t));
}
DensePolynomial interpolate_selector(const KZGProverKey& pk, const std::vector<Fr>& values) {
    if (values.empty()) throw Error(PLK_ERR_INVALID_ARG, "called `Option::unwrap()` on a `None` value");
    const Domain d = Domain::with_at_least(values.size());
    std::vector<Fr> out(d.size());
    uint32_t log_n = 0;
    check(pk.ctx(), plk_interpolate_selector(pk.ctx(), &values[0].v[0], values.size(), &out[0].v[0], &log_n));
    return DensePolynomial::from_coefficients_vec(std::move(out));
}
std::pair<DensePolynomial, DensePolynomial> interpolate_permutation_polynomials(const KZGProverKey& pk, const std::vector<uint64_t>& sigma, const Domain& domain) {
    if (sigma.size() > domain.size()) throw Error(PLK_ERR_INVALID_ARG, "Sigma length must match domain size");   // src/poly_utils.rs:22
    std::vector<Fr> s_id(domain.size()), s_sigma(domain.size());
    check(pk.ctx(), plk_interpolate_permutation(pk.ctx(), sigma.data(), sigma.size(), domain.log_size, &s_id[0].v[0], &s_sigma[0].v[0]));
    return {DensePolynomial::from_coefficients_vec(std::move(s_id)), DensePolynomial::from_coefficients_vec(std::move(s_sigma))};
}

}  // namespace plonk
