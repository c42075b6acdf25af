// Verifier of the reference (src/verifier.rs:57-236) restated on the host: replays the Fiat-Shamir transcript and runs
// PCS::check for the nine openings.  Like the reference it checks only the openings, never the gate or permutation identity.
//
// PCS::check for one polynomial without degree bound (ark-poly-commit 0.5 marlin_pc::check -> kzg10::check), with the opening
// challenge c that the prover's sponge produced:
//     e(c C - (c v) G - random_v gamma G, H) == e(W, beta H - zeta H)
// The pairing is the ate pairing f_{T,Q}(P)^((q^12 - 1)/r) with T = t - 1 = 6 u^2 on BN254 (G1 = E(Fq): y^2 = x^3 + 3, G2 on the
// sextic D-twist over Fq2 = Fq[i]/(i^2 + 1)).  The check compares two pairing values for equality, for which any bilinear
// non-degenerate pairing gives the same verdict as arkworks' optimal ate pairing.  Simplicity over speed: affine line
// functions, dense Fq12 products, the final exponentiation by plain square-and-multiply; a proof verifies in about 0.2 s.
// All of this runs on the CPU (pairings are O(1) work per proof and out of the GPU's scope).
#include "plonk_host.hpp"
#include "bn254_ext.hpp"
#include <string.h>

namespace plonk {

namespace {

using namespace ext;

// ---------------------------------------------------------------------------------------- tower Fq6 / Fq12 over Fq2
struct F6 { F2 c0, c1, c2; };                 // c0 + c1 v + c2 v^2,  v^3 = xi = 9 + i
struct F12 { F6 a, b; };                      // a + b w,  w^2 = v

F6 f6_zero() { return {f2_zero(), f2_zero(), f2_zero()}; }
F6 f6_one() { return {f2_one(), f2_zero(), f2_zero()}; }
F6 f6_add(const F6& x, const F6& y) { return {f2_add(x.c0, y.c0), f2_add(x.c1, y.c1), f2_add(x.c2, y.c2)}; }
F6 f6_sub(const F6& x, const F6& y) { return {f2_sub(x.c0, y.c0), f2_sub(x.c1, y.c1), f2_sub(x.c2, y.c2)}; }
F6 f6_neg(const F6& x) { return {f2_neg(x.c0), f2_neg(x.c1), f2_neg(x.c2)}; }
F6 f6_mul(const F6& x, const F6& y) {
    const F2 r0 = f2_add(f2_mul(x.c0, y.c0), f2_mul_xi(f2_add(f2_mul(x.c1, y.c2), f2_mul(x.c2, y.c1))));
    const F2 r1 = f2_add(f2_add(f2_mul(x.c0, y.c1), f2_mul(x.c1, y.c0)), f2_mul_xi(f2_mul(x.c2, y.c2)));
    const F2 r2 = f2_add(f2_add(f2_mul(x.c0, y.c2), f2_mul(x.c1, y.c1)), f2_mul(x.c2, y.c0));
    return {r0, r1, r2};
}
F6 f6_mul_v(const F6& x) { return {f2_mul_xi(x.c2), x.c0, x.c1}; }
F6 f6_inv(const F6& x) {
    const F2 t0 = f2_sub(f2_mul(x.c0, x.c0), f2_mul_xi(f2_mul(x.c1, x.c2)));
    const F2 t1 = f2_sub(f2_mul_xi(f2_mul(x.c2, x.c2)), f2_mul(x.c0, x.c1));
    const F2 t2 = f2_sub(f2_mul(x.c1, x.c1), f2_mul(x.c0, x.c2));
    const F2 d = f2_inv(f2_add(f2_mul(x.c0, t0), f2_mul_xi(f2_add(f2_mul(x.c2, t1), f2_mul(x.c1, t2)))));
    return {f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d)};
}

F12 f12_one() { return {f6_one(), f6_zero()}; }
F12 f12_mul(const F12& x, const F12& y) {
    const F6 aa = f6_mul(x.a, y.a), bb = f6_mul(x.b, y.b);
    const F6 cross = f6_mul(f6_add(x.a, x.b), f6_add(y.a, y.b));
    return {f6_add(aa, f6_mul_v(bb)), f6_sub(f6_sub(cross, aa), bb)};
}
F12 f12_conj(const F12& x) { return {x.a, f6_neg(x.b)}; }     // the q^6-power Frobenius
F12 f12_inv(const F12& x) {
    const F6 d = f6_inv(f6_sub(f6_mul(x.a, x.a), f6_mul_v(f6_mul(x.b, x.b))));
    return {f6_mul(x.a, d), f6_neg(f6_mul(x.b, d))};
}
bool f12_eq(const F12& x, const F12& y) {
    const F2* p = &x.a.c0; const F2* r = &y.a.c0;
    for (int i = 0; i < 6; ++i) if (!f2_eq(p[i], r[i])) return false;
    return true;
}
F12 f12_pow(const F12& x, const uint64_t* e, int words) {
    F12 r = f12_one();
    bool started = false;
    for (int i = 64 * words - 1; i >= 0; --i) {
        if (started) r = f12_mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) { r = started ? f12_mul(r, x) : x; started = true; }
    }
    return r;
}

// ---------------------------------------------------------------------------------------- pairing
struct G1 { el x, y; bool inf; };
// The untwist sends (x', y') to (x' w^2, y' w^3).  The line through T and R (slope l on the twist), evaluated at P:
//   yP - (l xP) w + (l xT - yT) w^3,   w^3 = v w   ->   a = (yP, 0, 0),  b = (-l xP, l xT - yT, 0)
F12 line_at(const G2& t, const F2& l, const G1& p) {
    F12 r{f6_zero(), f6_zero()};
    r.a.c0 = {p.y, hm::el_zero()};
    r.b.c0 = f2_neg(f2_mul_fq(l, p.x));
    r.b.c1 = f2_sub(f2_mul(l, t.x), t.y);
    return r;
}
// f_{T,Q}(P) for T = 6 u^2 = trace - 1 (127 bits)
F12 miller_loop(const G1& p, const G2& q) {
    if (p.inf || q.inf) return f12_one();
    static const uint64_t T[2] = {0xf83e9682e87cfd46ULL, 0x6f4d8248eeb859fbULL};
    F12 f = f12_one();
    G2 t = q;
    for (int i = 125; i >= 0; --i) {           // bit 126 is the leading one
        const F2 l = g2_slope(t, t);
        f = f12_mul(f12_mul(f, f), line_at(t, l, p));
        t = g2_add(t, t);
        if ((T[i / 64] >> (i % 64)) & 1) {
            const F2 l2 = g2_slope(t, q);
            f = f12_mul(f, line_at(t, l2, p));
            t = g2_add(t, q);
        }
    }
    return f;
}
F12 final_exponentiation(const F12& f) {
    static const uint64_t E1[8] = {0x3b5458a2275d69b2ULL, 0xa602072d09eac101ULL, 0x4a50189c6d96cadcULL, 0x04689e957a1242c8ULL,
                                   0x26edfa5c34c6b38dULL, 0xb00b855116375606ULL, 0x599a6f7c0348d21cULL, 0x0925c4b8763cbf9cULL};   // q^2 + 1
    static const uint64_t H[12] = {0xe81bb482ccdf42b1ULL, 0x5abf5cc4f49c36d4ULL, 0xf1154e7e1da014fdULL, 0xdcc7b44c87cdbacfULL,
                                   0xaaa441e3954bcf8aULL, 0x6b887d56d5095f23ULL, 0x79581e16f3fd90c6ULL, 0x3b1b1355d189227dULL,
                                   0x4e529a5861876f6bULL, 0x6c0eb522d5b12278ULL, 0x331ec15183177fafULL, 0x01baaa710b0759adULL};   // (q^4 - q^2 + 1) / r
    const F12 a = f12_mul(f12_conj(f), f12_inv(f));                       // f^(q^6 - 1)
    return f12_pow(f12_pow(a, E1, 8), H, 12);
}

G1 g1_of(const G1Affine& p) { return {hm::el_load(p.xy), hm::el_load(p.xy + 4), p.infinity != 0}; }
hm::pt pt_of(const G1Affine& p) { return p.infinity ? hm::pt_inf() : hm::pt_from_affine(p.xy); }
hm::pt pt_mul(const hm::pt& p, const el& k_mont) {
    uint64_t k[4];
    hm::to_canonical(FR, k_mont, k);
    hm::pt acc = hm::pt_inf();
    for (int i = 255; i >= 0; --i) {
        acc = hm::pt_dbl(acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = hm::pt_add(acc, p);
    }
    return acc;
}
hm::pt pt_neg(const hm::pt& p) { hm::pt r = p; r.y = hm::neg(FQ, p.y); return r; }
G1 g1_affine(const hm::pt& p) {
    uint64_t xy[8]; int32_t inf;
    hm::pt_to_affine(p, xy, &inf);
    return {hm::el_load(xy), hm::el_load(xy + 4), inf != 0};
}
G2 g2_of(const G2Affine& p) { return {{hm::el_load(p.x), hm::el_load(p.x + 4)}, {hm::el_load(p.y), hm::el_load(p.y + 4)}, p.infinity != 0}; }
G2Affine g2_store(const G2& p) {
    G2Affine r;
    memset(&r, 0, sizeof(r));
    r.infinity = p.inf ? 1 : 0;
    if (!p.inf) { memcpy(r.x, p.x.a.v, 32); memcpy(r.x + 4, p.x.b.v, 32); memcpy(r.y, p.y.a.v, 32); memcpy(r.y + 4, p.y.b.v, 32); }
    return r;
}

// kzg10::check: e(comm - v g - random_v gamma_g, h) == e(w, beta_h - zeta h), with the marlin_pc challenge folded in
bool kzg_check(const KZGVerifierKey& vk, const G2& beta_minus_zeta_h, const G2& h, const KZGCommitment& comm, const Fr& value, const KZGOpeningProof& proof,
               const el& challenge) {
    const el v = hm::mul(FR, challenge, hm::el_load(value.v));
    hm::pt inner = pt_mul(pt_of(comm.comm), challenge);
    inner = hm::pt_add(inner, pt_neg(pt_mul(pt_of(vk.g), v)));
    inner = hm::pt_add(inner, pt_neg(pt_mul(pt_of(vk.gamma_g), hm::el_load(proof.random_v.v))));
    // e(inner, h) * e(-w, beta_h - zeta h) == 1: two Miller loops, one final exponentiation
    const F12 f = f12_mul(miller_loop(g1_affine(inner), h), miller_loop(g1_affine(pt_neg(pt_of(proof.w))), beta_minus_zeta_h));
    return f12_eq(final_exponentiation(f), f12_one());
}

// inverse of serialize_g1_compressed: 32 bytes little endian x, 0x80 = "y is the larger of the two roots", 0x40 = infinity
G1Affine deserialize_g1_compressed(const uint8_t in[32]) {
    G1Affine p;
    memset(&p, 0, sizeof(p));
    uint8_t b[32];
    memcpy(b, in, 32);
    const uint8_t flags = b[31] & 0xc0;
    b[31] &= 0x3f;
    if (flags & 0x40) { p.infinity = 1; return p; }
    uint64_t xc[4];
    memcpy(xc, b, 32);
    if (hm::geq(xc, FQ.p)) throw Error(PLK_ERR_INVALID_ARG, "invalid point encoding: x is not reduced");
    const el x = hm::from_canonical(FQ, xc);
    const el rhs = hm::add(FQ, hm::mul(FQ, hm::mul(FQ, x, x), x), fq(3));
    static const uint64_t SQRT_EXP[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};   // (q + 1) / 4
    static const uint64_t HALF[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};       // (q - 1) / 2
    el y = hm::pow(FQ, rhs, SQRT_EXP);
    if (!hm::eq(hm::mul(FQ, y, y), rhs)) throw Error(PLK_ERR_INVALID_ARG, "invalid point encoding: x is not on the curve");
    uint64_t yc[4];
    hm::to_canonical(FQ, y, yc);
    bool larger = false;                                                   // y > (q - 1) / 2
    for (int i = 3; i >= 0; --i) if (yc[i] != HALF[i]) { larger = yc[i] > HALF[i]; break; }
    if (larger != ((flags & 0x80) != 0)) y = hm::neg(FQ, y);
    memcpy(p.xy, x.v, 32);
    memcpy(p.xy + 4, y.v, 32);
    return p;
}
Fr deserialize_fr(const uint8_t in[32]) {
    uint64_t c[4];
    memcpy(c, in, 32);
    if (hm::geq(c, FR.p)) throw Error(PLK_ERR_INVALID_ARG, "invalid field element encoding");
    const el e = hm::from_canonical(FR, c);
    Fr r;
    memcpy(r.v, e.v, 32);
    return r;
}

}  // namespace

PlonkProof deserialize_proof(const uint8_t* bytes, size_t len) {
    if (len != 1170) throw Error(PLK_ERR_INVALID_ARG, "a serialized proof has 1170 bytes");
    PlonkProof p;
    memset(&p, 0, sizeof(p));
    Fr* evals = &p.eval_a;
    KZGCommitment* comms = &p.comm_a;
    KZGOpeningProof* opens = &p.proof_a;
    const uint8_t* at = bytes;
    for (int i = 0; i < 9; ++i, at += 32) evals[i] = deserialize_fr(at);
    for (int i = 0; i < 9; ++i, at += 33) {
        comms[i].comm = deserialize_g1_compressed(at);
        if (at[32] != 0) throw Error(PLK_ERR_INVALID_ARG, "shifted_comm must be None");
    }
    for (int i = 0; i < 9; ++i, at += 65) {
        opens[i].w = deserialize_g1_compressed(at);
        if (at[32] != 1) throw Error(PLK_ERR_INVALID_ARG, "random_v must be Some");
        opens[i].random_v = deserialize_fr(at + 33);
    }
    return p;
}

KZGVerifierKey verifier_key_from_trapdoor(const Fr& tau, const Fr& gscal) {
    KZGVerifierKey vk;
    memset(&vk, 0, sizeof(vk));
    const el one = hm::one(FQ), two = fq(2);
    memcpy(vk.g.xy, one.v, 32);
    memcpy(vk.g.xy + 4, two.v, 32);
    const G1 gg = g1_affine(pt_mul(pt_of(vk.g), hm::el_load(gscal.v)));
    memcpy(vk.gamma_g.xy, gg.x.v, 32);
    memcpy(vk.gamma_g.xy + 4, gg.y.v, 32);
    vk.gamma_g.infinity = gg.inf ? 1 : 0;
    const G2 h = g2_generator();
    if (!g2_on_curve(h)) throw Error(PLK_ERR_INVALID_ARG, "G2 generator is not on the twist");
    vk.h = g2_store(h);
    vk.beta_h = g2_store(g2_mul(h, hm::el_load(tau.v)));
    return vk;
}

bool verify_plonk_proof_with_transcript(const KZGVerifierKey& vk, const PlonkProof& proof, const std::vector<Fr>& public_inputs, const Fr& opening_challenge) {
    // transcript replay, src/verifier.rs:100-167 (same absorb order as the prover)
    PlonkTranscript transcript("plonk_proof");
    transcript.absorb_public_inputs(public_inputs);
    transcript.absorb_commitment("q_add", proof.comm_q_add);
    transcript.absorb_commitment("q_mul", proof.comm_q_mul);
    transcript.absorb_commitment("a", proof.comm_a);
    transcript.absorb_commitment("b", proof.comm_b);
    transcript.absorb_commitment("c", proof.comm_c);
    transcript.absorb_commitment("s_id", proof.comm_s_id);
    transcript.absorb_commitment("s_sigma", proof.comm_s_sigma);
    (void)transcript.challenge_beta();
    (void)transcript.challenge_gamma();
    transcript.absorb_commitment("z", proof.comm_z);
    (void)transcript.challenge_alpha();
    transcript.absorb_commitment("t", proof.comm_t);
    const Fr zeta = transcript.challenge_zeta();
    transcript.absorb_evaluations(proof.eval_a, proof.eval_b, proof.eval_c, proof.eval_q_add, proof.eval_q_mul, proof.eval_s_id, proof.eval_s_sigma,
                                  proof.eval_z, proof.eval_t);
    // nine PCS::check calls, src/verifier.rs:205-213; all are evaluated, the results are and-ed (:225-226)
    const G2 h = g2_of(vk.h);
    const G2 bz = g2_add(g2_of(vk.beta_h), g2_neg(g2_mul(h, hm::el_load(zeta.v))));
    const KZGCommitment* comms = &proof.comm_a;
    const KZGOpeningProof* opens = &proof.proof_a;
    const Fr* evals = &proof.eval_a;
    bool ok = true;
    for (int i = 0; i < 9; ++i) ok = kzg_check(vk, bz, h, comms[i], evals[i], opens[i], hm::el_load(opening_challenge.v)) && ok;
    return ok;
}

// e(a G1, b G2) == e(G1, G2)^(a b) and e(G1, G2) != 1, for the tests
bool pairing_selftest(const Fr& a, const Fr& b) {
    G1Affine g;
    memset(&g, 0, sizeof(g));
    const el one = hm::one(FQ), two = fq(2);
    memcpy(g.xy, one.v, 32);
    memcpy(g.xy + 4, two.v, 32);
    const G2 h = g2_generator();
    const F12 base = final_exponentiation(miller_loop(g1_of(g), h));
    if (f12_eq(base, f12_one())) return false;
    const F12 lhs = final_exponentiation(miller_loop(g1_affine(pt_mul(pt_of(g), hm::el_load(a.v))), g2_mul(h, hm::el_load(b.v))));
    uint64_t ab[4];
    hm::to_canonical(FR, hm::mul(FR, hm::el_load(a.v), hm::el_load(b.v)), ab);
    return f12_eq(lhs, f12_pow(base, ab, 4));
}

}  // namespace plonk
