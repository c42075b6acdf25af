// BN254 G1 (y^2 = x^3 + 3) in extended Jacobian "XYZZ" coordinates over fp29.cuh's lazily reduced Fq.
//
// x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2.  The identity is ZZ = 0 (all limbs zero).
// Invariants of a stored point: X, Y, ZZ, ZZZ normalized; bound(X) <= 8, bound(Y) <= 4,
// bound(ZZ), bound(ZZZ) <= 2.  Coordinates carry the Montgomery factor 2^261.
// Affine operands (table entries) are canonical (bound 1), y possibly replaced by 2p - y (bound 2).
#pragma once
#include "fp29.cuh"

namespace plk {

struct xyzz {
    fe x, y, zz, zzz;
};
struct aff {
    fe x, y;
};

PLK_HD xyzz xyzz_inf() {
    xyzz r;
    r.x = fe_zero(); r.y = fe_zero(); r.zz = fe_zero(); r.zzz = fe_zero();
    return r;
}
PLK_HD bool xyzz_is_inf(const xyzz& p) { return fe_is_zero_limbs(p.zz); }

PLK_HD xyzz xyzz_from_affine(const aff& q) {
    xyzz r;
    r.x = q.x; r.y = q.y; r.zz = fe_const(FqP::ONE); r.zzz = fe_const(FqP::ONE);
    return r;
}
// -y as 2p - y for a canonical y: bound 2, limbs < 2^31 (not normalized; fine as a multiplication operand)
PLK_HD fe fq_neg_y(const fe& y) { return sub2<FqP>(fe_zero(), y); }

// 2 * (x, y) for an affine point, y != 0 on this curve (no points of order two)
PLK_HD xyzz xyzz_dbl_affine(const aff& q) {
    fe u = add(q.y, q.y);                         // bound 4, limbs < 2^30
    fe v = sqr<FqP>(u), w = mul<FqP>(u, v), s = mul<FqP>(q.x, v);
    fe xx = sqr<FqP>(q.x);
    fe m = add(add(xx, xx), xx); norm(m);         // bound 6
    fe s2 = add(s, s); norm(s2);                  // bound 4
    xyzz r;
    r.x = sub4<FqP>(sqr<FqP>(m), s2); norm(r.x);  // bound 6
    fe t = sub8<FqP>(s, r.x);                     // bound 10, limbs < 2^29 + 2^30
    r.y = sub2<FqP>(mul<FqP>(m, t), mul<FqP>(w, q.y)); norm(r.y);   // bound 4
    r.zz = v; r.zzz = w;
    return r;
}
PLK_HD xyzz xyzz_dbl(const xyzz& p) {
    if (xyzz_is_inf(p)) return p;
    fe u = add(p.y, p.y); norm(u);                // bound 8
    fe v = sqr<FqP>(u), w = mul<FqP>(u, v), s = mul<FqP>(p.x, v);
    fe xx = sqr<FqP>(p.x);                        // 64 <= 169
    fe m = add(add(xx, xx), xx); norm(m);         // bound 6
    fe s2 = add(s, s); norm(s2);
    xyzz r;
    r.x = sub4<FqP>(sqr<FqP>(m), s2); norm(r.x);  // bound 6
    fe t = sub8<FqP>(s, r.x);
    r.y = sub2<FqP>(mul<FqP>(m, t), mul<FqP>(w, p.y)); norm(r.y);   // bound 4
    r.zz = mul<FqP>(v, p.zz); r.zzz = mul<FqP>(w, p.zzz);
    return r;
}

// acc += (q.x, q.y)   (8M + 2S).  The general formula is evaluated whatever the operands are, and the two cases it does not cover
// (acc is the identity; acc and q have the same x) replace its result afterwards, in a branch that is almost never taken.  With the
// cases tested first and the general formula in the last arm of the branch, the compiler joined the arms with a register copy per
// limb of the result on the common path (38 moves per addition in the accumulation kernel's loop).
PLK_HD void xyzz_madd(xyzz& a, const aff& q) {
    const bool was_inf = xyzz_is_inf(a);
    fe u2 = mul<FqP>(q.x, a.zz), s2 = mul<FqP>(q.y, a.zzz);
    fe p = sub8<FqP>(u2, a.x); norm(p);           // bound 10
    fe r = sub4<FqP>(s2, a.y); norm(r);           // bound 6
    const bool same_x = is_zero_mod<FqP>(p);
    fe pp = sqr<FqP>(p), ppp = mul<FqP>(p, pp), qv = mul<FqP>(a.x, pp), rr = sqr<FqP>(r);
    fe x3;
#pragma unroll
    for (int i = 0; i < 9; ++i) x3.v[i] = rr.v[i] + FqP::SUB6X3[i] - ppp.v[i] - 2 * qv.v[i];
    norm(x3);                                     // bound 8
    fe t = sub8<FqP>(qv, x3);                     // bound 10, limbs < 3 * 2^29
    fe ny = sub4<FqP>(fe_zero(), a.y);            // 4p - Y: bound 4, limbs < 2^30
    fe y3 = mul2<FqP>(r, t, ny, ppp);             // r t - Y ppp in one reduction (60 + 8 <= 169): bound 2
    a.zz = mul<FqP>(a.zz, pp); a.zzz = mul<FqP>(a.zzz, ppp);
    a.x = x3; a.y = y3;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_expect(was_inf || same_x, 0)) {
#else
    if (was_inf || same_x) {
#endif
        if (was_inf) { a = xyzz_from_affine(q); norm(a.y); }                       // q.y may be the unnormalized 2p - y
        else if (is_zero_mod<FqP>(r)) { aff qq = q; norm(qq.y); a = xyzz_dbl_affine(qq); }
        else a = xyzz_inf();
    }
}

// a + b, both XYZZ (12M + 2S on the common path)
PLK_HD xyzz xyzz_add(const xyzz& a, const xyzz& b) {
    if (xyzz_is_inf(a)) return b;
    if (xyzz_is_inf(b)) return a;
    fe u1 = mul<FqP>(a.x, b.zz), u2 = mul<FqP>(b.x, a.zz), s1 = mul<FqP>(a.y, b.zzz), s2 = mul<FqP>(b.y, a.zzz);
    fe p = sub2<FqP>(u2, u1); norm(p);            // bound 4
    fe r = sub2<FqP>(s2, s1); norm(r);            // bound 4
    if (is_zero_mod<FqP>(p)) {
        if (is_zero_mod<FqP>(r)) return xyzz_dbl(a);
        return xyzz_inf();
    }
    fe pp = sqr<FqP>(p), ppp = mul<FqP>(p, pp), qv = mul<FqP>(u1, pp), rr = sqr<FqP>(r);
    xyzz o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.x.v[i] = rr.v[i] + FqP::SUB6X3[i] - ppp.v[i] - 2 * qv.v[i];
    norm(o.x);                                    // bound 8
    fe t = sub8<FqP>(qv, o.x);                    // bound 10, limbs < 3 * 2^29
    fe ns1 = sub2<FqP>(fe_zero(), s1);            // 2p - S1: bound 2, limbs < 2^30
    o.y = mul2<FqP>(r, t, ns1, ppp);              // r t - S1 ppp in one reduction (40 + 4 <= 169): bound 2
    o.zz = mul<FqP>(mul<FqP>(a.zz, b.zz), pp);
    o.zzz = mul<FqP>(mul<FqP>(a.zzz, b.zzz), ppp);
    return o;
}

// table entry in memory: x then y, each eight words, canonical, Montgomery factor 2^261
PLK_HD aff load_aff(const uint32_t* p) {
    aff q;
    q.x = load_fe(p);
    q.y = load_fe(p + 8);
    return q;
}

}  // namespace plk
