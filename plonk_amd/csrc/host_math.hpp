// Host-side BN254 arithmetic used by libplonk_hip for the short serial tails that a GPU lane is bad at:
// the last few dozen point additions of an MSM and its single inversion, O(1) scalar bookkeeping of a
// KZG opening, constants such as n^-1 and powers of zeta.  4x64-bit limbs, Montgomery radix 2^256 (the
// ABI's own representation).
#pragma once
#include <stdint.h>
#include <string.h>

namespace plk {
namespace host {

typedef unsigned __int128 u128;

struct Params {
    uint64_t p[4], one[4], r2[4], inv;
};
static const Params FR = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                          {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL},
                          {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL},
                          0xc2e1f593efffffffULL};
static const Params FQ = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
                          {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL},
                          {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL},
                          0x87d20782e4866389ULL};

struct el {
    uint64_t v[4];
};
inline el el_zero() { el r; memset(r.v, 0, 32); return r; }
inline el el_load(const uint64_t* p) { el r; memcpy(r.v, p, 32); return r; }
inline bool is_zero(const el& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline bool eq(const el& a, const el& b) { return memcmp(a.v, b.v, 32) == 0; }

inline bool geq(const uint64_t* a, const uint64_t* b) {
    for (int i = 3; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
inline void sub_inplace(uint64_t* a, const uint64_t* b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
}
inline el add(const Params& f, const el& a, const el& b) {
    el r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    if (geq(r.v, f.p)) sub_inplace(r.v, f.p);
    return r;
}
inline el sub(const Params& f, const el& a, const el& b) {
    el r = a; uint64_t br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.v[i] + f.p[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
inline el neg(const Params& f, const el& a) { return is_zero(a) ? a : sub(f, el_zero(), a); }
// Montgomery product; accepts any a < 2^256 as long as b < p
inline el mul(const Params& f, const el& a, const el& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * f.inv;
        c = ((u128)m * f.p[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * f.p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    el r; memcpy(r.v, t, 32);
    if (t[4] || geq(r.v, f.p)) sub_inplace(r.v, f.p);
    return r;
}
inline el one(const Params& f) { return el_load(f.one); }
inline el from_canonical(const Params& f, const uint64_t* c) { return mul(f, el_load(c), el_load(f.r2)); }
inline el from_u64(const Params& f, uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return from_canonical(f, c); }
inline void to_canonical(const Params& f, const el& a, uint64_t* out) {
    el o = el_zero(); o.v[0] = 1;
    el r = mul(f, a, o); memcpy(out, r.v, 32);
}
inline el pow(const Params& f, const el& a, const uint64_t* e) {
    el r = one(f);
    for (int i = 255; i >= 0; --i) { r = mul(f, r, r); if ((e[i / 64] >> (i % 64)) & 1) r = mul(f, r, a); }
    return r;
}
inline el inv(const Params& f, const el& a) {
    uint64_t e[4]; memcpy(e, f.p, 32); e[0] -= 2;
    return pow(f, a, e);
}

// ---- G1 in XYZZ coordinates over FQ -----------------------------------------------------------
struct pt {
    el x, y, zz, zzz;   // identity: zz == 0
};
inline pt pt_inf() { pt r; r.x = r.y = r.zz = r.zzz = el_zero(); return r; }
inline pt pt_load(const uint64_t* p) { pt r; r.x = el_load(p); r.y = el_load(p + 4); r.zz = el_load(p + 8); r.zzz = el_load(p + 12); return r; }
inline void pt_store(const pt& a, uint64_t* p) { memcpy(p, a.x.v, 32); memcpy(p + 4, a.y.v, 32); memcpy(p + 8, a.zz.v, 32); memcpy(p + 12, a.zzz.v, 32); }
inline pt pt_dbl(const pt& a) {
    if (is_zero(a.zz)) return a;
    const Params& f = FQ;
    el u = add(f, a.y, a.y), v = mul(f, u, u), w = mul(f, u, v), s = mul(f, a.x, v);
    el xx = mul(f, a.x, a.x), m = add(f, add(f, xx, xx), xx);
    pt r;
    r.x = sub(f, mul(f, m, m), add(f, s, s));
    r.y = sub(f, mul(f, m, sub(f, s, r.x)), mul(f, w, a.y));
    r.zz = mul(f, v, a.zz); r.zzz = mul(f, w, a.zzz);
    return r;
}
inline pt pt_add(const pt& a, const pt& b) {
    if (is_zero(a.zz)) return b;
    if (is_zero(b.zz)) return a;
    const Params& f = FQ;
    el u1 = mul(f, a.x, b.zz), u2 = mul(f, b.x, a.zz), s1 = mul(f, a.y, b.zzz), s2 = mul(f, b.y, a.zzz);
    el p = sub(f, u2, u1), r = sub(f, s2, s1);
    if (is_zero(p)) return is_zero(r) ? pt_dbl(a) : pt_inf();
    el pp = mul(f, p, p), ppp = mul(f, p, pp), q = mul(f, u1, pp);
    pt o;
    o.x = sub(f, sub(f, mul(f, r, r), ppp), add(f, q, q));
    o.y = sub(f, mul(f, r, sub(f, q, o.x)), mul(f, s1, ppp));
    o.zz = mul(f, mul(f, a.zz, b.zz), pp); o.zzz = mul(f, mul(f, a.zzz, b.zzz), ppp);
    return o;
}
// -> affine x||y (8 limbs) + infinity flag
inline void pt_to_affine(const pt& a, uint64_t* xy, int32_t* is_inf) {
    if (is_zero(a.zz)) { memset(xy, 0, 64); *is_inf = 1; return; }
    const Params& f = FQ;
    el i = inv(f, mul(f, a.zz, a.zzz));
    el x = mul(f, a.x, mul(f, i, a.zzz)), y = mul(f, a.y, mul(f, i, a.zz));
    memcpy(xy, x.v, 32); memcpy(xy + 4, y.v, 32); *is_inf = 0;
}
inline pt pt_from_affine(const uint64_t* xy) { pt r; r.x = el_load(xy); r.y = el_load(xy + 4); r.zz = one(FQ); r.zzz = one(FQ); return r; }

}  // namespace host
}  // namespace plk
