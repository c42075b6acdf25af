// merlin 3.0 transcript (STROBE-128 over Keccak-f[1600]) and ark-serialize's compressed encodings, host side.
// Mirrors src/transcript.rs of the reference; the Fiat-Shamir transcript stays on the CPU by design.
#include "plonk_host.hpp"
#include "../csrc/host_math.hpp"
#include <string.h>

namespace plonk {

namespace {

inline uint64_t rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

// Keccak-f[1600] on a little-endian byte state, written lane by lane (x + 5y)
void keccak_f(uint8_t* bytes) {
    static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int RHO[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
    uint64_t a[5][5];   // [x][y]
    for (int y = 0; y < 5; ++y)
        for (int x = 0; x < 5; ++x) memcpy(&a[x][y], bytes + 8 * (x + 5 * y), 8);
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[5][5];
        for (int x = 0; x < 5; ++x) c[x] = a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y][(2 * x + 3 * y) % 5] = rotl(a[x][y] ^ d[x], RHO[x][y]);
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) a[x][y] = b[x][y] ^ (~b[(x + 1) % 5][y] & b[(x + 2) % 5][y]);
        a[0][0] ^= RC[round];
    }
    for (int y = 0; y < 5; ++y)
        for (int x = 0; x < 5; ++x) memcpy(bytes + 8 * (x + 5 * y), &a[x][y], 8);
}

constexpr uint8_t STROBE_R = 166;
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16, FLAG_K = 32;

}  // namespace

PlonkTranscript::PlonkTranscript(const char* domain_separator) {
    memset(st_, 0, sizeof(st_));
    const uint8_t header[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(st_, header, 6);
    memcpy(st_ + 6, "STROBEv1.0.2", 12);
    keccak_f(st_);
    meta_ad(reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11, false);
    append_message("dom-sep", reinterpret_cast<const uint8_t*>(domain_separator), strlen(domain_separator));
}
void PlonkTranscript::run_f() {
    st_[pos_] ^= pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[STROBE_R + 1] ^= 0x80;
    keccak_f(st_);
    pos_ = 0;
    pos_begin_ = 0;
}
void PlonkTranscript::absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        st_[pos_++] ^= d[i];
        if (pos_ == STROBE_R) run_f();
    }
}
void PlonkTranscript::squeeze(uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        d[i] = st_[pos_];
        st_[pos_++] = 0;
        if (pos_ == STROBE_R) run_f();
    }
}
void PlonkTranscript::begin_op(uint8_t flags, bool more) {
    if (more) return;   // continuation of the current operation
    const uint8_t old_begin = pos_begin_;
    pos_begin_ = pos_ + 1;
    cur_flags_ = flags;
    const uint8_t frame[2] = {old_begin, flags};
    absorb(frame, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
}
void PlonkTranscript::meta_ad(const uint8_t* d, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(d, n);
}
void PlonkTranscript::append_message(const char* label, const uint8_t* msg, size_t len) {
    const uint32_t l = (uint32_t)len;
    uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
    meta_ad(reinterpret_cast<const uint8_t*>(label), strlen(label), false);
    meta_ad(le, 4, true);
    begin_op(FLAG_A, false);
    absorb(msg, len);
}
void PlonkTranscript::challenge_bytes(const char* label, uint8_t* out, size_t len) {
    const uint32_t l = (uint32_t)len;
    uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
    meta_ad(reinterpret_cast<const uint8_t*>(label), strlen(label), false);
    meta_ad(le, 4, true);
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    squeeze(out, len);
}

// src/transcript.rs:14-18: 64 bytes, little endian, reduced mod r
Fr PlonkTranscript::challenge_scalar(const char* label) {
    using namespace plk::host;
    uint8_t buf[64];
    challenge_bytes(label, buf, 64);
    uint64_t lo[4], hi[4];
    memcpy(lo, buf, 32);
    memcpy(hi, buf + 32, 32);
    // lo + hi * 2^256: from_canonical reduces a 256-bit input and yields x*2^256; one more product by 2^512 shifts hi up by 2^256
    const el l = from_canonical(FR, lo);
    const el h = mul(FR, from_canonical(FR, hi), el_load(FR.r2));
    const el s = add(FR, l, h);
    Fr out;
    memcpy(out.v, s.v, 32);
    return out;
}

void serialize_fr(const Fr& x, uint8_t out[32]) {
    uint64_t c[4];
    plk::host::to_canonical(plk::host::FR, plk::host::el_load(x.v), c);
    memcpy(out, c, 32);
}
// SW compressed: x little endian; bit 7 of the last byte set when y > -y, bit 6 for the point at infinity
void serialize_g1_compressed(const G1Affine& p, uint8_t out[32]) {
    using namespace plk::host;
    memset(out, 0, 32);
    if (p.infinity) { out[31] |= 0x40; return; }
    uint64_t x[4], y[4], ny[4];
    to_canonical(FQ, el_load(p.xy), x);
    const el ey = el_load(p.xy + 4);
    to_canonical(FQ, ey, y);
    to_canonical(FQ, neg(FQ, ey), ny);
    memcpy(out, x, 32);
    bool y_is_larger = false;
    for (int i = 3; i >= 0; --i)
        if (y[i] != ny[i]) { y_is_larger = y[i] > ny[i]; break; }
    if (y_is_larger) out[31] |= 0x80;
}

void PlonkTranscript::absorb_field(const char* label, const Fr& x) {
    uint8_t b[32];
    serialize_fr(x, b);
    append_message(label, b, 32);
}
void PlonkTranscript::absorb_commitment(const char* label, const KZGCommitment& c) {
    uint8_t b[33];
    serialize_g1_compressed(c.comm, b);
    b[32] = 0;   // shifted_comm: None
    append_message(label, b, 33);
}
void PlonkTranscript::absorb_public_inputs(const std::vector<Fr>& public_inputs) {
    if (public_inputs.empty()) return;   // src/transcript.rs:49
    std::vector<uint8_t> buf(32 * public_inputs.size());
    for (size_t i = 0; i < public_inputs.size(); ++i) serialize_fr(public_inputs[i], &buf[32 * i]);
    append_message("public_inputs", buf.data(), buf.size());
}
void PlonkTranscript::absorb_evaluations(const Fr& a, const Fr& b, const Fr& c, const Fr& q_add, const Fr& q_mul, const Fr& s_id, const Fr& s_sigma, const Fr& z,
                                         const Fr& t) {
    absorb_field("eval_a", a);
    absorb_field("eval_b", b);
    absorb_field("eval_c", c);
    absorb_field("eval_q_add", q_add);
    absorb_field("eval_q_mul", q_mul);
    absorb_field("eval_s_id", s_id);
    absorb_field("eval_s_sigma", s_sigma);
    absorb_field("eval_z", z);
    absorb_field("eval_t", t);
}

std::vector<uint8_t> serialize_proof(const PlonkProof& p) {
    std::vector<uint8_t> out(9 * 32 + 9 * 33 + 9 * 65, 0);
    uint8_t* o = out.data();
    const Fr* ev = &p.eval_a;
    for (int i = 0; i < 9; ++i) { serialize_fr(ev[i], o); o += 32; }
    const KZGCommitment* cm = &p.comm_a;
    for (int i = 0; i < 9; ++i) { serialize_g1_compressed(cm[i].comm, o); o[32] = 0; o += 33; }
    const KZGOpeningProof* pr = &p.proof_a;
    for (int i = 0; i < 9; ++i) { serialize_g1_compressed(pr[i].w, o); o[32] = 1; serialize_fr(pr[i].random_v, o + 33); o += 65; }
    return out;
}

}  // namespace plonk
