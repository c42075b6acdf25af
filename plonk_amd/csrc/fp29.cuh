// BN254 field arithmetic for gfx950: nine 29-bit limbs, Montgomery radix 2^261.
//
// Why 29-bit limbs: on MI355X v_mad_u64_u32 issues at the same rate as a 64-bit add, so the cost of
// a modular product is its instruction count.  With 29-bit limbs every column of the schoolbook
// product (at most 9 + 9 partial products of < 2^60) fits one 64-bit accumulator: a column is a
// pure chain of v_mad_u64_u32 with no carry, compare or move instructions in between
// (171 multiply-adds per product, against >500 instructions for 8x32-bit limbs).
//
// Representation.  fe.v[i] holds limb i (little endian, weight 2^(29 i)); limb 8 holds everything
// from bit 232 upwards.  Values are kept lazily reduced; each routine states what it needs:
//   "normalized" : limbs 0..7 < 2^29
//   "bound k"    : value < k * p
// mul() needs  limb(a) * limb(b) <= 2^60.6 for every pair of limbs  and  bound(a) * bound(b) <= 169,
// and returns a normalized value with bound 2.
//
// Montgomery scale.  mul(a, b) = a * b * 2^-261 mod p.  Memory formats keep the usual arkworks form
// x * 2^256 (eight 32-bit words); kernels pick their constants so the scale comes out right:
//   mul(x*2^256, w*2^261) = x*w*2^256     (data times library constant keeps the data's form)
//   mul(x*2^256, TO_RP)   = x*2^261       mul(x*2^261, TO_R) = x*2^256
//   mul(x*2^256, RAW32)   = x             mul(x*2^261, RAW1) = x
#pragma once
#include <stdint.h>
#include "constants29.h"

#if defined(__HIPCC__)
#define PLK_HD __host__ __device__ __forceinline__
#else
#define PLK_HD inline
#endif

namespace plk {

constexpr uint32_t M29 = 0x1fffffffu;

struct fe {
    uint32_t v[9];
};

PLK_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = 0;
    return r;
}
PLK_HD fe fe_const(const uint32_t (&c)[9]) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = c[i];
    return r;
}
PLK_HD bool fe_is_zero_limbs(const fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= a.v[i];
    return o == 0;
}

// eight 32-bit words <-> nine 29-bit limbs (bit re-slicing only, no change of value)
PLK_HD fe unpack(const uint32_t* w) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, j = bit / 32, o = bit % 32;
        uint32_t x = w[j] >> o;
        if (o + 29 > 32 && j + 1 < 8) x |= w[j + 1] << (32 - o);
        r.v[i] = (i < 8) ? (x & M29) : x;
    }
    return r;
}
// requires normalized limbs and value < 2^256
PLK_HD void pack(const fe& a, uint32_t* w) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, i = bit / 29, o = bit % 29;
        uint32_t x = a.v[i] >> o;
        x |= a.v[i + 1] << (29 - o);
        w[j] = x;
    }
}

// acc + a * b, one v_mad_u64_u32.  On the device every partial sum carries an assumption (true: the operand requirements of the
// routines below keep a column below 2^64 - 2^56).  It generates no code, but gives the partial sum a second use, and the compiler's
// reassociation pass only reorders sums whose intermediate values have a single use: without it the pass moves the carry that
// starts a column to the end of the column's sum (it is the operand that becomes available last), and every column then costs a
// 64-bit addition on top of its multiply-adds (17 per product).  Inline assembly per multiply-add has the same effect on the sum but
// makes the compiler put a wait state after every statement, which costs more than the additions.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t mad_vv(uint32_t a, uint32_t b, uint64_t c) {
    c += (uint64_t)a * b;
    __builtin_assume((uint32_t)(c >> 32) != 0xffffffffu);
    return c;
}
#else
PLK_HD uint64_t mad_vv(uint32_t a, uint32_t b, uint64_t c) { return c + (uint64_t)a * b; }
#endif
PLK_HD uint64_t mad_vs(uint32_t a, uint32_t b, uint64_t c) { return mad_vv(a, b, c); }   // b: a limb of the modulus

// a * b * 2^-261 mod p (see the header comment for operand requirements); result normalized, bound 2
template <class F>
PLK_HD fe mul(const fe& a, const fe& b) {
    uint64_t acc = 0;
    uint32_t m[9];
    fe r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vv(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        m[k] = ((uint32_t)acc * F::NINV) & M29;
        acc = mad_vs(m[k], F::P[0], acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vv(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

// (a * b + c * d) * 2^-261 mod p with a single Montgomery reduction; result normalized, bound 2.
// Needs  9 * (limb(a) * limb(b) + limb(c) * limb(d)) + 9 * 2^58 < 2^64  for the column accumulators
// (for instance limbs a < 2^29, b < 3 * 2^29, c < 2^30, d < 2^29)  and  bound(a) * bound(b) + bound(c) * bound(d) <= 169.
template <class F>
PLK_HD fe mul2(const fe& a, const fe& b, const fe& c, const fe& d) {
    uint64_t acc = 0;
    uint32_t m[9];
    fe r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vv(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vv(c.v[i], d.v[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        m[k] = ((uint32_t)acc * F::NINV) & M29;
        acc = mad_vs(m[k], F::P[0], acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vv(a.v[i], b.v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vv(c.v[i], d.v[k - i], acc);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

// a * a * 2^-261 mod p; cross products are computed once against the doubled operand
template <class F>
PLK_HD fe sqr(const fe& a) {
    uint64_t acc = 0;
    uint32_t m[9], d[9];
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = a.v[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc = mad_vv(d[i], a.v[k - i], acc);
        if ((k & 1) == 0) acc = mad_vv(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
        for (int i = 0; i < k; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        m[k] = ((uint32_t)acc * F::NINV) & M29;
        acc = mad_vs(m[k], F::P[0], acc);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc = mad_vv(d[i], a.v[k - i], acc);
        if ((k & 1) == 0) acc = mad_vv(a.v[k / 2], a.v[k / 2], acc);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc = mad_vs(m[i], F::P[k - i], acc);
        r.v[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}

// limb-wise sum, no carry: limb and value bounds add
PLK_HD fe add(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// carry propagation: afterwards limbs 0..7 < 2^29 (value unchanged)
PLK_HD void norm(fe& a) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a.v[i + 1] += a.v[i] >> 29;
        a.v[i] &= M29;
    }
}
// a - b + k*p, limb-wise against a k*p whose low limbs are >= 2^29 - 1.
// Needs b normalized with bound k; the result is NOT normalized (limbs < limb(a) + 2^30), bound = bound(a) + k.
#define PLK_DEFINE_SUB(K)                                              \
    template <class F>                                                 \
    PLK_HD fe sub##K(const fe& a, const fe& b) {                       \
        fe r;                                                          \
        _Pragma("unroll") for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + F::SUB##K[i] - b.v[i]; \
        return r;                                                      \
    }
PLK_DEFINE_SUB(2)
PLK_DEFINE_SUB(4)
PLK_DEFINE_SUB(8)
PLK_DEFINE_SUB(16)
PLK_DEFINE_SUB(32)
#undef PLK_DEFINE_SUB

// normalized value with bound 2  ->  canonical representative in [0, p)
template <class F>
PLK_HD fe reduce(const fe& a) {
    fe d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t t = a.v[i] - F::P[i] - borrow;
        borrow = t >> 31;
        d.v[i] = (i < 8) ? (t & M29) : t;
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = borrow ? a.v[i] : d.v[i];
    return r;
}

// ---- product with a fixed operand (Shoup) ---------------------------------------------------------------------------------
// For a factor w that is used many times (a twiddle), the quotient of x * w by p can be estimated from a precomputed
// wq = floor(w * 2^261 / p):  q = floor(x * wq / 2^261)  is at most 2 below floor(x * w / p), so  r = x * w - q * p  lies in [0, 3p).
// There is no chain of reduction multipliers (each Montgomery column waits for the multiplier of the column before) and no
// multiply by -p^-1: 53 multiply-adds for the upper columns of x * wq (columns 0..6 only matter through their carry, which is
// below 2^-23 of a unit of q), 45 for the low nine columns of x * w, 45 for those of q * (2^261 - p); r is their sum modulo 2^261.
// The product is a plain one, x * w mod p, with no Montgomery factor: data in the x * 2^256 form keeps its form.
//
// Needs  x < 2^261 (bound 169)  and  limb(x) <= 2.5 * 2^30  (18 partial products of a column stay below 2^64);  w canonical, wq < 2^261,
// both normalized.  Returns a normalized value with bound 3.
struct shoup {
    fe w, wq;
};
template <class F>
PLK_HD fe mul_shoup(const fe& x, const shoup& c) {
    uint64_t acc = 0;
    uint32_t q[9];
    // upper columns of x * wq: columns 7 and 8 feed the carry only
#pragma unroll
    for (int k = 7; k < 17; ++k) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc = mad_vv(x.v[i], c.wq.v[k - i], acc);
        if (k >= 9) q[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    q[8] = (uint32_t)acc;
    // low nine columns of x * w + q * (2^261 - p)
    fe r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vv(x.v[i], c.w.v[k - i], acc);
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vs(q[i], F::NEGP[k - i], acc);
        r.v[k] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    return r;   // r < 3p < 2^256: limb 8 is below 2^24, nothing is lost above it
}
// The pair (w, wq) from w * 2^261 mod p, canonical (the form the twiddle tables hold):  w * 2^261 = wq * p + s  with
// s = w * 2^261 mod p, so  wq = -s * p^-1 mod 2^261  (the division is exact), the low half of one product.
template <class F>
PLK_HD shoup shoup_of(const fe& s) {
    shoup c;
    c.w = reduce<F>(mul<F>(s, fe_const(F::RAW1)));
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc = mad_vs(s.v[i], F::NPINV[k - i], acc);
        c.wq.v[k] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    return c;
}
// (p - w, its quotient factor) from (w, wq), w != 0:  (p - w) 2^261 = (2^261 - 1 - wq) p + (p - s)
template <class F>
PLK_HD shoup shoup_neg(const shoup& c) {
    shoup n;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint32_t t = F::P[i] - c.w.v[i] - borrow;
        borrow = t >> 31;
        n.w.v[i] = t & M29;
        n.wq.v[i] = M29 - c.wq.v[i];
    }
    return n;
}
// normalized value with bound 3  ->  canonical representative in [0, p)
template <class F>
PLK_HD fe reduce3(const fe& a) { return reduce<F>(reduce<F>(a)); }

// exact test "a == 0 (mod p)" for a normalized value with bound 64.
// a = k*p forces k = a[0] * p^-1 mod 2^29, so almost every non-multiple is rejected by one multiply.
template <class F>
PLK_HD bool is_zero_mod(const fe& a) {
    const uint32_t k = (a.v[0] * F::PINV) & M29;
    if (k >= 64) return false;
    uint64_t c = 0;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c += (uint64_t)k * F::P[i];
        const uint32_t limb = (i < 8) ? ((uint32_t)c & M29) : (uint32_t)c;
        eq = eq && (limb == a.v[i]);
        c >>= 29;
    }
    return eq;
}

// a^(p-2): inverse in the 2^261 Montgomery form (a = x*2^261 -> x^-1 * 2^261); a normalized, bound 2
template <class F>
PLK_HD fe inv(const fe& a) {
    // exponent p - 2 as four 64-bit words, reassembled from the 29-bit limbs of p
    uint64_t e[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i;
        e[bit / 64] |= (uint64_t)F::P[i] << (bit % 64);
        if (bit % 64 > 35 && bit / 64 + 1 < 4) e[bit / 64 + 1] |= (uint64_t)F::P[i] >> (64 - bit % 64);
    }
    e[0] -= 2;
    fe r = fe_const(F::ONE);
    for (int i = 253; i >= 0; --i) {
        r = sqr<F>(r);
        const uint64_t w = (i >= 192) ? e[3] : (i >= 128) ? e[2] : (i >= 64) ? e[1] : e[0];
        if ((w >> (i & 63)) & 1) r = mul<F>(r, a);
    }
    return r;
}

// memory (x*2^256, eight words, canonical) -> limbs, same value
PLK_HD fe load_fe(const uint32_t* p) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = p[i];
    return unpack(w);
}
// normalized value with bound 2 -> canonical eight words
template <class F>
PLK_HD void store_fe(const fe& a, uint32_t* p) {
    uint32_t w[8];
    pack(reduce<F>(a), w);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = w[i];
}

}  // namespace plk
