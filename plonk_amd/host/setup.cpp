// The randomness of the reference's driver restated on the host, so that a run of src/main.rs can be replayed end to end:
//   ark_std::test_rng()           src/main.rs:25, src/prover.rs:48    -> StdRng, test_rng
//   Fr::rand(rng)                 src/main.rs:136-137,174-176         -> fr_rand
//   PCS::setup + PCS::trim        src/main.rs:182-183                 -> KZGProverKey::setup (powers computed on the GPU)
//   test_sponge + the challenge PCS::open squeezes from it            src/prover.rs:36-59,285-303 -> test_sponge_challenge
// The algorithms are those of the crates the reference pins (rand 0.8 / rand_chacha 0.3 through ark-std 0.5; ark-ff, ark-ec,
// ark-poly-commit, ark-crypto-primitives 0.5).  test_rng() is deterministic in ark-std 0.5 only when the environment variable
// DETERMINISTIC_TEST_RNG is 1; that is the branch restated here (without it the reference's prover and verifier sponges draw
// different round constants and its own verification fails).
#include "plonk_host.hpp"
#include "bn254_ext.hpp"
#include <string.h>

namespace plonk {

using namespace ext;

// ------------------------------------------------------------------------------------------------ StdRng = ChaCha12Rng
namespace {
inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
inline void quarter(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
}
}  // namespace

void chacha_block(const uint8_t key[32], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 8; ++i)
        init[4 + i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    init[12] = (uint32_t)counter; init[13] = (uint32_t)(counter >> 32);
    init[14] = (uint32_t)stream; init[15] = (uint32_t)(stream >> 32);
    uint32_t w[16];
    memcpy(w, init, sizeof w);
    for (int r = 0; r < rounds / 2; ++r) {
        quarter(w, 0, 4, 8, 12); quarter(w, 1, 5, 9, 13); quarter(w, 2, 6, 10, 14); quarter(w, 3, 7, 11, 15);
        quarter(w, 0, 5, 10, 15); quarter(w, 1, 6, 11, 12); quarter(w, 2, 7, 8, 13); quarter(w, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = w[i] + init[i];
}

StdRng::StdRng(const uint8_t seed[32]) { memcpy(key_, seed, 32); }
// rand_chacha produces four consecutive blocks per refill; rand_core's BlockRng hands them out word by word
void StdRng::refill(uint32_t index) {
    for (int b = 0; b < 4; ++b) chacha_block(key_, counter_++, 0, 12, buf_ + 16 * b);
    index_ = index;
}
uint32_t StdRng::next_u32() {
    if (index_ >= 64) refill(0);
    return buf_[index_++];
}
uint64_t StdRng::next_u64() {
    const uint32_t i = index_;
    if (i < 63) { index_ += 2; return (uint64_t)buf_[i] | ((uint64_t)buf_[i + 1] << 32); }
    if (i >= 64) { refill(2); return (uint64_t)buf_[0] | ((uint64_t)buf_[1] << 32); }
    const uint64_t lo = buf_[63];      // one word left: it becomes the low half, the high half comes from the next buffer
    refill(1);
    return lo | ((uint64_t)buf_[0] << 32);
}
bool StdRng::gen_bool() { return (next_u32() >> 31) != 0; }

StdRng test_rng() {
    static const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0};
    return StdRng(seed);
}

// ------------------------------------------------------------------------------------------------ field and group sampling
namespace {
// ark-ff `Distribution<Fp> for Standard`: four limbs, the two unused top bits cleared, rejection; the limbs ARE the Montgomery form
el fp_rand(StdRng& rng, const hm::Params& f) {
    for (;;) {
        el r;
        for (int i = 0; i < 4; ++i) r.v[i] = rng.next_u64();
        r.v[3] &= ~0ull >> 2;
        if (!hm::geq(r.v, f.p)) return r;
    }
}
// canonical integers compared as ark-ff's `Ord for Fp` does
bool fq_less(const el& a, const el& b) {
    uint64_t x[4], y[4];
    hm::to_canonical(FQ, a, x); hm::to_canonical(FQ, b, y);
    for (int i = 3; i >= 0; --i) if (x[i] != y[i]) return x[i] < y[i];
    return false;
}
// q = 3 mod 4: a^((q + 1) / 4), checked
bool fq_sqrt(const el& a, el* out) {
    static const uint64_t E[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};   // (q + 1) / 4
    const el r = hm::pow(FQ, a, E);
    if (!hm::eq(hm::mul(FQ, r, r), a)) return false;
    *out = r;
    return true;
}
bool fq_is_qr(const el& a) {      // Legendre symbol +1 (zero is neither residue nor non-residue)
    static const uint64_t E[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};   // (q - 1) / 2
    return hm::eq(hm::pow(FQ, a, E), hm::one(FQ));
}
// ark-ff QuadExtField::sqrt (the "complex method"); callers order the two roots themselves
bool f2_sqrt(const F2& a, F2* out) {
    if (hm::is_zero(a.b)) {
        if (hm::is_zero(a.a)) { *out = f2_zero(); return true; }
        el r;
        if (fq_is_qr(a.a)) { if (!fq_sqrt(a.a, &r)) return false; *out = {r, hm::el_zero()}; return true; }
        if (!fq_sqrt(hm::neg(FQ, a.a), &r)) return false;          // c0 / nonresidue with nonresidue = -1
        *out = {hm::el_zero(), r};
        return true;
    }
    el alpha;
    if (!fq_sqrt(hm::add(FQ, hm::mul(FQ, a.a, a.a), hm::mul(FQ, a.b, a.b)), &alpha)) return false;   // root of the norm
    const el two_inv = hm::inv(FQ, fq(2));
    el delta = hm::mul(FQ, hm::add(FQ, alpha, a.a), two_inv);
    if (!hm::is_zero(delta) && !fq_is_qr(delta)) delta = hm::sub(FQ, delta, alpha);
    el c0;
    if (!fq_sqrt(delta, &c0) || hm::is_zero(c0)) return false;
    const F2 cand = {c0, hm::mul(FQ, hm::mul(FQ, a.b, two_inv), hm::inv(FQ, c0))};
    if (!f2_eq(f2_mul(cand, cand), a)) return false;
    *out = cand;
    return true;
}
// ark-ff's `Ord for QuadExtField`: by c1, then by c0
bool f2_less(const F2& x, const F2& y) {
    if (!hm::eq(x.b, y.b)) return fq_less(x.b, y.b);
    return fq_less(x.a, y.a);
}
G2Affine g2_store(const G2& p) {
    G2Affine o{};
    o.infinity = p.inf ? 1 : 0;
    if (!p.inf) { memcpy(o.x, p.x.a.v, 32); memcpy(o.x + 4, p.x.b.v, 32); memcpy(o.y, p.y.a.v, 32); memcpy(o.y + 4, p.y.b.v, 32); }
    return o;
}
}  // namespace

Fr fr_rand(StdRng& rng) {
    const el r = fp_rand(rng, FR);
    Fr o; memcpy(o.v, r.v, 32);
    return o;
}
Rng as_rng(StdRng& rng) { return [&rng]() { return fr_rand(rng); }; }

// ark-ec `Distribution<Projective<P>> for Standard`: x, then the bool that picks the larger or the smaller y; cofactor 1
G1Affine g1_rand(StdRng& rng) {
    for (;;) {
        const el x = fp_rand(rng, FQ);
        const bool greatest = rng.gen_bool();
        el y;
        if (!fq_sqrt(hm::add(FQ, hm::mul(FQ, hm::mul(FQ, x, x), x), fq(3)), &y)) continue;
        const el ny = hm::neg(FQ, y);
        const bool y_is_smaller = fq_less(y, ny);
        const el pick = (greatest == y_is_smaller) ? ny : y;
        G1Affine p{};
        memcpy(p.xy, x.v, 32); memcpy(p.xy + 4, pick.v, 32);
        return p;
    }
}
// the same on the twist, followed by the multiplication by the cofactor 2q - r
G2Affine g2_rand(StdRng& rng) {
    static const uint64_t COFACTOR[4] = {0x345f2299c0f9fa8dULL, 0x06ceecda572a2489ULL, 0xb85045b68181585eULL, 0x30644e72e131a029ULL};
    const F2 b = f2_mul_fq(f2_inv(f2_mul_xi(f2_one())), fq(3));
    for (;;) {
        F2 x;
        x.a = fp_rand(rng, FQ);
        x.b = fp_rand(rng, FQ);
        const bool greatest = rng.gen_bool();
        F2 y;
        if (!f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), b), &y)) continue;
        const F2 ny = f2_neg(y);
        const bool y_is_smaller = f2_less(y, ny);
        const F2 pick = (greatest == y_is_smaller) ? ny : y;
        return g2_store(g2_mul_int(G2{x, pick, false}, COFACTOR));
    }
}

// KZG10::setup draws beta, g, gamma_g, h in this order; MarlinKZG10::trim(pp, d, 1, None) keeps powers_of_g[0..=d] and three powers
// of gamma_g.  The d + 1 powers are computed on the device.
KZGVerifierKey KZGProverKey::setup_cyclic(uint64_t max_degree, StdRng& rng, uint32_t r, uint32_t w, uint32_t log_block) {
    if (max_degree < 1) throw Error(PLK_ERR_INVALID_ARG, "called `Result::unwrap()` on an `Err` value: DegreeIsZero");
    const Fr beta = fr_rand(rng);
    KZGVerifierKey vk{};
    vk.g = g1_rand(rng);
    vk.gamma_g = g1_rand(rng);
    vk.h = g2_rand(rng);
    const G2 h{{hm::el_load(vk.h.x), hm::el_load(vk.h.x + 4)}, {hm::el_load(vk.h.y), hm::el_load(vk.h.y + 4)}, false};
    vk.beta_h = g2_store(g2_mul(h, hm::el_load(beta.v)));
    const int32_t rc = plk_srs_setup_cyclic(ctx_, beta.v, vk.g.xy, vk.gamma_g.xy, r, w, log_block, max_degree + 1);
    if (rc != PLK_OK) throw Error(rc, std::string("libplonk_hip: ") + plk_last_error(ctx_));
    len_ = max_degree + 1;
    return vk;
}
KZGVerifierKey KZGProverKey::setup(uint64_t max_degree, StdRng& rng) { return setup_cyclic(max_degree, rng, 0, 1, 0); }

// ------------------------------------------------------------------------------------------------ test_sponge
// PoseidonConfig::new(8, 31, 17, mds, ark, rate 2, capacity 1) with 39 x 3 round constants drawn from a fresh test_rng().
// PCS::open asks a fresh sponge for one Truncated(128) element: the sponge permutes its zero state once, hands out the first rate
// element (state[capacity]) and the low 128 bits of its canonical integer become the challenge.
Fr test_sponge_challenge() {
    StdRng rng = test_rng();
    el ark[39][3];
    for (auto& row : ark) for (auto& c : row) c = fp_rand(rng, FR);
    el s[3] = {hm::el_zero(), hm::el_zero(), hm::el_zero()};
    auto pow17 = [](const el& x) {
        el y = x;
        for (int i = 0; i < 4; ++i) y = hm::mul(FR, y, y);
        return hm::mul(FR, y, x);
    };
    for (int rnd = 0; rnd < 39; ++rnd) {
        for (int i = 0; i < 3; ++i) s[i] = hm::add(FR, s[i], ark[rnd][i]);
        if (rnd < 4 || rnd >= 35) for (auto& x : s) x = pow17(x);
        else s[0] = pow17(s[0]);
        // MDS rows (1 0 1), (1 1 0), (0 1 1)
        const el t0 = hm::add(FR, s[0], s[2]), t1 = hm::add(FR, s[0], s[1]), t2 = hm::add(FR, s[1], s[2]);
        s[0] = t0; s[1] = t1; s[2] = t2;
    }
    uint64_t c[4];
    hm::to_canonical(FR, s[1], c);
    c[2] = c[3] = 0;
    const el m = hm::from_canonical(FR, c);
    Fr o; memcpy(o.v, m.v, 32);
    return o;
}

}  // namespace plonk
