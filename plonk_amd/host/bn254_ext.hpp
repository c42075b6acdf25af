// Fq2 = Fq[i]/(i^2 + 1) and the G2 group of BN254 (affine points on the sextic twist y^2 = x^3 + 3/(9 + i)) for the host-side
// verifier and the KZG setup.  Simplicity over speed: these run a handful of times per proof or per setup.
#pragma once
#include "../csrc/host_math.hpp"

namespace plonk {
namespace ext {

using plk::host::el;
using plk::host::FQ;
using plk::host::FR;
namespace hm = plk::host;

struct F2 { el a, b; };                       // a + b i

inline el fq(uint64_t x) { return hm::from_u64(FQ, x); }
inline F2 f2_zero() { return {hm::el_zero(), hm::el_zero()}; }
inline F2 f2_one() { return {hm::one(FQ), hm::el_zero()}; }
inline bool f2_is_zero(const F2& x) { return hm::is_zero(x.a) && hm::is_zero(x.b); }
inline bool f2_eq(const F2& x, const F2& y) { return hm::eq(x.a, y.a) && hm::eq(x.b, y.b); }
inline F2 f2_add(const F2& x, const F2& y) { return {hm::add(FQ, x.a, y.a), hm::add(FQ, x.b, y.b)}; }
inline F2 f2_sub(const F2& x, const F2& y) { return {hm::sub(FQ, x.a, y.a), hm::sub(FQ, x.b, y.b)}; }
inline F2 f2_neg(const F2& x) { return {hm::neg(FQ, x.a), hm::neg(FQ, x.b)}; }
inline F2 f2_mul(const F2& x, const F2& y) {
    const el aa = hm::mul(FQ, x.a, y.a), bb = hm::mul(FQ, x.b, y.b);
    const el cross = hm::mul(FQ, hm::add(FQ, x.a, x.b), hm::add(FQ, y.a, y.b));
    return {hm::sub(FQ, aa, bb), hm::sub(FQ, hm::sub(FQ, cross, aa), bb)};
}
inline F2 f2_mul_fq(const F2& x, const el& k) { return {hm::mul(FQ, x.a, k), hm::mul(FQ, x.b, k)}; }
inline F2 f2_inv(const F2& x) {
    const el d = hm::inv(FQ, hm::add(FQ, hm::mul(FQ, x.a, x.a), hm::mul(FQ, x.b, x.b)));
    return {hm::mul(FQ, x.a, d), hm::neg(FQ, hm::mul(FQ, x.b, d))};
}
inline F2 f2_mul_xi(const F2& x) {                   // (a + b i)(9 + i) = 9a - b + (9b + a) i
    const el nine = fq(9);
    return {hm::sub(FQ, hm::mul(FQ, x.a, nine), x.b), hm::add(FQ, hm::mul(FQ, x.b, nine), x.a)};
}

// ---------------------------------------------------------------------------------------- G2 (affine, on the twist y^2 = x^3 + 3/xi)
struct G2 { F2 x, y; bool inf; };
inline G2 g2_neg(const G2& p) { return {p.x, f2_neg(p.y), p.inf}; }
// slope of the line through p and r (tangent when they coincide); both finite, r != -p
inline F2 g2_slope(const G2& p, const G2& r) {
    if (f2_eq(p.x, r.x)) {
        const F2 xx = f2_mul(p.x, p.x);
        return f2_mul(f2_add(f2_add(xx, xx), xx), f2_inv(f2_add(p.y, p.y)));
    }
    return f2_mul(f2_sub(r.y, p.y), f2_inv(f2_sub(r.x, p.x)));
}
inline G2 g2_add(const G2& p, const G2& r) {
    if (p.inf) return r;
    if (r.inf) return p;
    if (f2_eq(p.x, r.x) && !f2_eq(p.y, r.y)) return {f2_zero(), f2_zero(), true};
    if (f2_eq(p.x, r.x) && f2_is_zero(p.y)) return {f2_zero(), f2_zero(), true};
    const F2 l = g2_slope(p, r);
    const F2 x3 = f2_sub(f2_sub(f2_mul(l, l), p.x), r.x);
    return {x3, f2_sub(f2_mul(l, f2_sub(p.x, x3)), p.y), false};
}
// k: 256-bit integer, little-endian words
inline G2 g2_mul_int(const G2& p, const uint64_t k[4]) {
    G2 acc{f2_zero(), f2_zero(), true};
    for (int i = 255; i >= 0; --i) {
        acc = g2_add(acc, acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = g2_add(acc, p);
    }
    return acc;
}
inline G2 g2_mul(const G2& p, const el& k_mont) {
    uint64_t k[4];
    hm::to_canonical(FR, k_mont, k);
    return g2_mul_int(p, k);
}
inline G2 g2_generator() {
    static const uint64_t X0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    static const uint64_t X1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    static const uint64_t Y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    static const uint64_t Y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    return {{hm::from_canonical(FQ, X0), hm::from_canonical(FQ, X1)}, {hm::from_canonical(FQ, Y0), hm::from_canonical(FQ, Y1)}, false};
}
inline bool g2_on_curve(const G2& p) {
    if (p.inf) return true;
    const F2 b = f2_mul_fq(f2_inv(f2_mul_xi(f2_one())), fq(3));          // 3 / xi
    return f2_eq(f2_mul(p.y, p.y), f2_add(f2_mul(f2_mul(p.x, p.x), p.x), b));
}


}  // namespace ext
}  // namespace plonk
