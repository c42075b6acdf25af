// CPU restatement of the reference prover hot path -- TEST INFRASTRUCTURE ONLY.
//
// Nothing under plonk_amd/ may link, load or call this file: it is the checker that the
// tests, the smoke test and bench.py's cpu_baseline leg compare the HIP library against.
// PARITY UNPINNED: the reference (sabrinahirani/PLONK) ships no tests, golden vectors or
// fixtures, and cannot be built here (Rust; its arithmetic lives in the un-vendored crates
// ark-ff / ark-ec / ark-poly / ark-poly-commit / ark-bn254 / ark-serialize 0.5 and merlin 3.0,
// Cargo.toml:9-18).  This file restates their published algorithms at the reference's call
// sites and is pinned by: the merlin known-answer test, the BN254 constants, the reference's
// example-circuit values, and an independent big-integer model (oracle/pymodel.py).
//
// Plain single-threaded C++ (the reference enables neither `parallel` nor `asm`):
//   4x64-bit-limb Montgomery Fr/Fq, Jacobian G1, radix-2 NTT, signed-window Pippenger,
//   hiding KZG10 commit/open, merlin STROBE-128, compressed serialization, and the
//   round-by-round flow of src/prover.rs:119-336.
//
// ABI used by the ctypes wrapper (oracle/oracle.py): field elements are 4 little-endian
// u64 limbs in Montgomery form; points are x||y (8 limbs) plus an infinity flag.
#include <cstdint>
#include <cstring>
#include <vector>
#include <chrono>
#include <cmath>
#include <stdexcept>

typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------- fields
struct FrTag {}; struct FqTag {};
template <class T> struct Mod;
template <> struct Mod<FrTag> {
    static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
};
template <> struct Mod<FqTag> {
    static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
    static constexpr uint64_t INV = 0x87d20782e4866389ULL;
};
constexpr uint64_t Mod<FrTag>::P[4]; constexpr uint64_t Mod<FrTag>::ONE[4]; constexpr uint64_t Mod<FrTag>::R2[4];
constexpr uint64_t Mod<FqTag>::P[4]; constexpr uint64_t Mod<FqTag>::ONE[4]; constexpr uint64_t Mod<FqTag>::R2[4];

template <class T> struct Fp {
    uint64_t v[4];
    static Fp zero() { Fp r; memset(r.v, 0, 32); return r; }
    static Fp one() { Fp r; memcpy(r.v, Mod<T>::ONE, 32); return r; }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, 32) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }
};

template <class T> static inline bool geq_p(const uint64_t* a) {
    for (int i = 3; i >= 0; --i) { if (a[i] != Mod<T>::P[i]) return a[i] > Mod<T>::P[i]; }
    return true;
}
template <class T> static inline void sub_p(uint64_t* a) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - Mod<T>::P[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
template <class T> static inline Fp<T> add(const Fp<T>& a, const Fp<T>& b) {
    Fp<T> r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    if (geq_p<T>(r.v)) sub_p<T>(r.v);   // p < 2^254: the sum never carries out of 256 bits
    return r;
}
template <class T> static inline Fp<T> sub(const Fp<T>& a, const Fp<T>& b) {
    Fp<T> r; u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a.v[i] - b.v[i] - br; r.v[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.v[i] + Mod<T>::P[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
template <class T> static inline Fp<T> neg(const Fp<T>& a) { return a.is_zero() ? a : sub(Fp<T>::zero(), a); }
template <class T> static inline Fp<T> dbl(const Fp<T>& a) { return add(a, a); }

// CIOS Montgomery product, R = 2^256.
template <class T> static inline Fp<T> mul(const Fp<T>& a, const Fp<T>& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * Mod<T>::INV;
        c = (u128)m * Mod<T>::P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * Mod<T>::P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fp<T> r; memcpy(r.v, t, 32);
    if (t[4] || geq_p<T>(r.v)) sub_p<T>(r.v);
    return r;
}
template <class T> static inline Fp<T> sqr(const Fp<T>& a) { return mul(a, a); }
template <class T> static Fp<T> from_canonical(const uint64_t* c) { Fp<T> x, r2; memcpy(x.v, c, 32); memcpy(r2.v, Mod<T>::R2, 32); return mul(x, r2); }
template <class T> static void to_canonical(const Fp<T>& a, uint64_t* out) {
    Fp<T> one; one.v[0] = 1; one.v[1] = one.v[2] = one.v[3] = 0;
    Fp<T> r = mul(a, one); memcpy(out, r.v, 32);
}
template <class T> static Fp<T> from_u64(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return from_canonical<T>(c); }
template <class T> static Fp<T> pow_limbs(const Fp<T>& a, const uint64_t* e) {
    Fp<T> r = Fp<T>::one();
    for (int i = 255; i >= 0; --i) { r = sqr(r); if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, a); }
    return r;
}
template <class T> static Fp<T> inv(const Fp<T>& a) {   // a^(p-2); inverse of zero is zero
    uint64_t e[4]; memcpy(e, Mod<T>::P, 32); e[0] -= 2;
    return pow_limbs(a, e);
}
template <class T> static void batch_inv(Fp<T>* a, size_t n) {   // Montgomery's trick; entries must be non-zero
    if (!n) return;
    std::vector<Fp<T>> pre(n);
    Fp<T> acc = Fp<T>::one();
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; acc = mul(acc, a[i]); }
    acc = inv(acc);
    for (size_t i = n; i-- > 0;) { Fp<T> t = mul(acc, pre[i]); acc = mul(acc, a[i]); a[i] = t; }
}
typedef Fp<FrTag> Fr; typedef Fp<FqTag> Fq;

// ------------------------------------------------------------------------------------- G1: y^2 = x^3 + 3 (Jacobian)
struct Aff { Fq x, y; bool inf; };
struct Jac { Fq x, y, z; bool is_inf() const { return z.is_zero(); } };
static Jac jac_inf() { Jac r; r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero(); return r; }
static Jac jac_dbl(const Jac& p) {
    if (p.is_inf()) return p;
    Fq A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    Fq D = dbl(sub(sub(sqr(add(p.x, B)), A), C));
    Fq E = add(dbl(A), A), F = sqr(E);
    Jac r; r.x = sub(F, dbl(D));
    Fq C8 = dbl(dbl(dbl(C)));
    r.z = dbl(mul(p.y, p.z));
    r.y = sub(mul(E, sub(D, r.x)), C8);
    return r;
}
static Jac jac_add_mixed(const Jac& p, const Aff& q) {
    if (q.inf) return p;
    if (p.is_inf()) { Jac r; r.x = q.x; r.y = q.y; r.z = Fq::one(); return r; }
    Fq Z1Z1 = sqr(p.z), U2 = mul(q.x, Z1Z1), S2 = mul(mul(q.y, p.z), Z1Z1);
    if (U2 == p.x) { if (S2 == p.y) return jac_dbl(p); return jac_inf(); }
    Fq H = sub(U2, p.x), HH = sqr(H), I = dbl(dbl(HH)), J = mul(H, I), r = dbl(sub(S2, p.y)), V = mul(p.x, I);
    Jac o; o.x = sub(sub(sqr(r), J), dbl(V));
    o.y = sub(mul(r, sub(V, o.x)), dbl(mul(p.y, J)));
    o.z = sub(sub(sqr(add(p.z, H)), Z1Z1), HH);
    return o;
}
static Jac jac_add(const Jac& p, const Jac& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    Fq Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z), U1 = mul(p.x, Z2Z2), U2 = mul(q.x, Z1Z1);
    Fq S1 = mul(mul(p.y, q.z), Z2Z2), S2 = mul(mul(q.y, p.z), Z1Z1);
    if (U1 == U2) { if (S1 == S2) return jac_dbl(p); return jac_inf(); }
    Fq H = sub(U2, U1), I = sqr(dbl(H)), J = mul(H, I), r = dbl(sub(S2, S1)), V = mul(U1, I);
    Jac o; o.x = sub(sub(sqr(r), J), dbl(V));
    o.y = sub(mul(r, sub(V, o.x)), dbl(mul(S1, J)));
    o.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return o;
}
static Aff to_affine(const Jac& p) {
    Aff a; a.inf = p.is_inf();
    if (a.inf) { a.x = Fq::zero(); a.y = Fq::zero(); return a; }
    Fq zi = inv(p.z), zi2 = sqr(zi);
    a.x = mul(p.x, zi2); a.y = mul(p.y, mul(zi2, zi));
    return a;
}
static Aff aff_neg(const Aff& a) { Aff r = a; r.y = neg(a.y); return r; }
static Jac jac_mul_canonical(const Aff& p, const uint64_t* k) {
    Jac r = jac_inf();
    for (int i = 255; i >= 0; --i) { r = jac_dbl(r); if ((k[i / 64] >> (i % 64)) & 1) r = jac_add_mixed(r, p); }
    return r;
}
static Aff load_aff(const uint64_t* xy, int is_inf) { Aff a; memcpy(a.x.v, xy, 32); memcpy(a.y.v, xy + 4, 32); a.inf = is_inf != 0; return a; }
static void store_aff(const Aff& a, uint64_t* xy, int32_t* is_inf) {
    if (a.inf) memset(xy, 0, 64); else { memcpy(xy, a.x.v, 32); memcpy(xy + 4, a.y.v, 32); }
    *is_inf = a.inf ? 1 : 0;
}

// Threads the MSM may use: 1 (the default) is the reference's arkworks build (no `parallel` feature, Cargo.toml:9-18); more is only
// for the all-cores figure of the CPU baseline (orc_set_threads).
static int g_threads = 1;
// Variable-base MSM: signed-digit windows + bucket method (ark-ec VariableBaseMSM, restated).
static Jac msm_pippenger(const Fr* scalars, const uint64_t* bases_xy, size_t n) {
    if (n == 0) return jac_inf();
    int c = n < 32 ? 3 : (int)std::floor(std::log((double)n)) + 2;   // window width ln(n) + 2
    if (c > 20) c = 20;
    const int nwin = (254 + c - 1) / c + 1;
    const size_t half = (size_t)1 << (c - 1);
    std::vector<int32_t> digits(n * nwin);
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[5]; to_canonical(scalars[i], k); k[4] = 0;
        int carry = 0;
        for (int w = 0; w < nwin; ++w) {
            int bit = w * c; int64_t d = 0;
            if (bit < 256) {
                unsigned limb = bit / 64, off = bit % 64;
                u128 two = (u128)k[limb] | ((u128)k[limb + 1] << 64);
                d = (int64_t)((uint64_t)(two >> off) & (((uint64_t)1 << c) - 1));
            }
            d += carry;
            if (d > (int64_t)half) { d -= (int64_t)1 << c; carry = 1; } else carry = 0;
            digits[i * nwin + w] = (int32_t)d;
        }
    }
    std::vector<Jac> wsum(nwin);
    if (g_threads > 1 && n >= 4096) {
        // All-cores form for the CPU baseline: (window, point range) tasks with a bucket array each, dealt to the threads; the
        // partial window sums of the ranges are added afterwards.  Same digits, same buckets, same result as the serial loop below.
        const int nchunk = std::max(1, (4 * g_threads + nwin - 1) / nwin);
        const size_t per = (n + nchunk - 1) / nchunk;
        std::vector<Jac> part((size_t)nwin * nchunk);
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
        for (int task = 0; task < nwin * nchunk; ++task) {
            const int w = task / nchunk, ch = task % nchunk;
            const size_t lo = (size_t)ch * per, hi = std::min(n, lo + per);
            std::vector<Jac> bk(half, jac_inf());
            for (size_t i = lo; i < hi; ++i) {
                int32_t d = digits[i * nwin + w];
                if (d == 0) continue;
                Aff p = load_aff(bases_xy + 8 * i, 0);
                if (d > 0) bk[d - 1] = jac_add_mixed(bk[d - 1], p);
                else bk[-d - 1] = jac_add_mixed(bk[-d - 1], aff_neg(p));
            }
            Jac run = jac_inf(), acc = jac_inf();
            for (size_t b = half; b-- > 0;) { run = jac_add(run, bk[b]); acc = jac_add(acc, run); }
            part[task] = acc;
        }
        for (int w = 0; w < nwin; ++w) {
            Jac acc = jac_inf();
            for (int ch = 0; ch < nchunk; ++ch) acc = jac_add(acc, part[(size_t)w * nchunk + ch]);
            wsum[w] = acc;
        }
        Jac total = wsum[nwin - 1];
        for (int w = nwin - 2; w >= 0; --w) { for (int j = 0; j < c; ++j) total = jac_dbl(total); total = jac_add(total, wsum[w]); }
        return total;
    }
    std::vector<Jac> buckets(half);
    for (int w = 0; w < nwin; ++w) {
        for (auto& b : buckets) b = jac_inf();
        for (size_t i = 0; i < n; ++i) {
            int32_t d = digits[i * nwin + w];
            if (d == 0) continue;
            Aff p = load_aff(bases_xy + 8 * i, 0);
            if (d > 0) buckets[d - 1] = jac_add_mixed(buckets[d - 1], p);
            else buckets[-d - 1] = jac_add_mixed(buckets[-d - 1], aff_neg(p));
        }
        Jac run = jac_inf(), acc = jac_inf();
        for (size_t b = half; b-- > 0;) { run = jac_add(run, buckets[b]); acc = jac_add(acc, run); }
        wsum[w] = acc;
    }
    Jac total = wsum[nwin - 1];
    for (int w = nwin - 2; w >= 0; --w) { for (int j = 0; j < c; ++j) total = jac_dbl(total); total = jac_add(total, wsum[w]); }
    return total;
}

// ------------------------------------------------------------------------------------- NTT (ark-poly Radix2EvaluationDomain semantics)
static Fr root_of_unity(int log_n) {
    static const uint64_t ROOT28[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};
    Fr w; memcpy(w.v, ROOT28, 32);
    for (int i = log_n; i < 28; ++i) w = sqr(w);
    return w;
}
// natural order in and out; inverse multiplies by n^-1; coset g: forward evaluates at g*w^i, inverse undoes it.
static void ntt_inplace(Fr* a, int log_n, bool inverse, const Fr* coset) {
    const size_t n = (size_t)1 << log_n;
    if (coset && !inverse) { Fr g = Fr::one(); for (size_t i = 0; i < n; ++i) { a[i] = mul(a[i], g); g = mul(g, *coset); } }
    for (size_t i = 0; i < n; ++i) {
        size_t j = 0; for (int b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) std::swap(a[i], a[j]);
    }
    Fr w = root_of_unity(log_n); if (inverse) w = inv(w);
    std::vector<Fr> tw(n / 2 ? n / 2 : 1); tw[0] = Fr::one();
    for (size_t i = 1; i < n / 2; ++i) tw[i] = mul(tw[i - 1], w);
    for (int s = 1; s <= log_n; ++s) {
        size_t m = (size_t)1 << s, step = n / m;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < m / 2; ++j) {
                Fr t = mul(a[k + j + m / 2], tw[j * step]);
                Fr u = a[k + j];
                a[k + j] = add(u, t); a[k + j + m / 2] = sub(u, t);
            }
    }
    if (inverse) {
        Fr ninv = inv(from_u64<FrTag>(n));
        if (coset) { Fr gi = inv(*coset), g = ninv; for (size_t i = 0; i < n; ++i) { a[i] = mul(a[i], g); g = mul(g, gi); } }
        else for (size_t i = 0; i < n; ++i) a[i] = mul(a[i], ninv);
    }
}

// ------------------------------------------------------------------------------------- dense polynomials
static size_t trimmed_len(const Fr* c, size_t len) { while (len && c[len - 1].is_zero()) --len; return len; }
static Fr horner(const Fr* c, size_t len, const Fr& x) { Fr acc = Fr::zero(); for (size_t i = len; i-- > 0;) acc = add(mul(acc, x), c[i]); return acc; }
// quotient of c(X)/(X - z), remainder dropped; out has len-1 entries
static void div_linear(const Fr* c, size_t len, const Fr& z, Fr* out) {
    Fr carry = Fr::zero();
    for (size_t i = len - 1; i > 0; --i) { carry = add(c[i], mul(carry, z)); out[i - 1] = carry; }
}

// ------------------------------------------------------------------------------------- merlin 3.0 (STROBE-128 over Keccak-f[1600])
static void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int r = 0; r < 24; ++r) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; ++i) { uint64_t t = bc[(i + 4) % 5] ^ ((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63)); for (int j = 0; j < 25; j += 5) s[j + i] ^= t; }
        uint64_t t = s[1];
        for (int i = 0; i < 24; ++i) { int j = PILN[i]; uint64_t b = s[j]; s[j] = (t << ROTC[i]) | (t >> (64 - ROTC[i])); t = b; }
        for (int j = 0; j < 25; j += 5) { for (int i = 0; i < 5; ++i) bc[i] = s[j + i]; for (int i = 0; i < 5; ++i) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5]; }
        s[0] ^= RC[r];
    }
}
struct Transcript {
    enum { R = 166, FI = 1, FA = 2, FC = 4, FT = 8, FM = 16, FK = 32 };
    union { uint64_t w[25]; uint8_t b[200]; } st;
    uint8_t pos, pos_begin, cur_flags;
    explicit Transcript(const char* label) {
        memset(st.b, 0, 200);
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(st.b, hdr, 6); memcpy(st.b + 6, "STROBEv1.0.2", 12);
        keccak_f1600(st.w); pos = pos_begin = cur_flags = 0;
        meta_ad((const uint8_t*)"Merlin v1.0", 11, false);
        append_message("dom-sep", (const uint8_t*)label, strlen(label));
    }
    void run_f() { st.b[pos] ^= pos_begin; st.b[pos + 1] ^= 0x04; st.b[R + 1] ^= 0x80; keccak_f1600(st.w); pos = 0; pos_begin = 0; }
    void absorb(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; ++i) { st.b[pos++] ^= d[i]; if (pos == R) run_f(); } }
    void squeeze(uint8_t* d, size_t n) { for (size_t i = 0; i < n; ++i) { d[i] = st.b[pos]; st.b[pos++] = 0; if (pos == R) run_f(); } }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t old = pos_begin; pos_begin = pos + 1; cur_flags = flags;
        uint8_t d[2] = {old, flags}; absorb(d, 2);
        if ((flags & (FC | FK)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FM | FA, more); absorb(d, n); }
    void ad(const uint8_t* d, size_t n, bool more) { begin_op(FA, more); absorb(d, n); }
    void prf(uint8_t* d, size_t n) { begin_op(FI | FA | FC, false); squeeze(d, n); }
    void append_message(const char* label, const uint8_t* m, size_t n) {
        uint32_t len = (uint32_t)n; meta_ad((const uint8_t*)label, strlen(label), false); meta_ad((const uint8_t*)&len, 4, true); ad(m, n, false);
    }
    void challenge_bytes(const char* label, uint8_t* out, size_t n) {
        uint32_t len = (uint32_t)n; meta_ad((const uint8_t*)label, strlen(label), false); meta_ad((const uint8_t*)&len, 4, true); prf(out, n);
    }
    // src/transcript.rs:14-18: 64 bytes little-endian mod r
    Fr challenge_scalar(const char* label) {
        uint8_t buf[64]; challenge_bytes(label, buf, 64);
        uint64_t lo[4], hi[4]; memcpy(lo, buf, 32); memcpy(hi, buf + 32, 32);
        // value = lo + hi*2^256; with Montgomery form x -> x*R: mont(lo) = lo*R, and hi*2^256 = hi*R, whose Montgomery form is hi*R^2
        Fr l = from_canonical<FrTag>(lo);                 // reduces any 256-bit input mod r
        Fr r2; memcpy(r2.v, Mod<FrTag>::R2, 32);
        Fr h = mul(from_canonical<FrTag>(hi), r2);
        return add(l, h);
    }
};

// ------------------------------------------------------------------------------------- serialization (ark-serialize compressed)
static void fr_bytes(const Fr& x, uint8_t* out) { uint64_t c[4]; to_canonical(x, c); memcpy(out, c, 32); }
static void g1_compressed(const Aff& p, uint8_t* out) {
    if (p.inf) { memset(out, 0, 32); out[31] |= 0x40; return; }
    uint64_t x[4], y[4], ny[4]; to_canonical(p.x, x); to_canonical(p.y, y); to_canonical(neg(p.y), ny);
    memcpy(out, x, 32);
    bool bigger = false;
    for (int i = 3; i >= 0; --i) if (y[i] != ny[i]) { bigger = y[i] > ny[i]; break; }
    if (bigger) out[31] |= 0x80;
}

// ------------------------------------------------------------------------------------- KZG10 with hiding bound 1 (via marlin_pc)
struct Srs { const uint64_t* g; size_t ng; const uint64_t* gg; };
static Aff kzg_commit(const Srs& s, const Fr* coeffs, size_t len, const Fr* blind) {
    len = trimmed_len(coeffs, len);
    size_t lead = 0; while (lead < len && coeffs[lead].is_zero()) ++lead;
    if (len > s.ng) throw std::runtime_error("polynomial longer than the SRS");
    Jac c = msm_pippenger(coeffs + lead, s.g + 8 * lead, len - lead);
    size_t bl = trimmed_len(blind, 3);
    Jac h = msm_pippenger(blind, s.gg, bl);
    return to_affine(jac_add(c, h));
}
static void kzg_open(const Srs& s, const Fr* coeffs, size_t len, const Fr* blind, const Fr& zeta, const Fr& chal, Aff* W, Fr* random_v) {
    len = trimmed_len(coeffs, len);
    std::vector<Fr> p(len), w(len ? len - 1 : 0);
    for (size_t i = 0; i < len; ++i) p[i] = mul(coeffs[i], chal);
    size_t plen = trimmed_len(p.data(), len);
    Jac acc = jac_inf();
    if (plen >= 2) {
        div_linear(p.data(), plen, zeta, w.data());
        size_t wl = trimmed_len(w.data(), plen - 1), lead = 0;
        while (lead < wl && w[lead].is_zero()) ++lead;
        acc = msm_pippenger(w.data() + lead, s.g + 8 * lead, wl - lead);
    }
    Fr r[3], hw[2];
    for (int i = 0; i < 3; ++i) r[i] = mul(blind[i], chal);
    size_t rl = trimmed_len(r, 3);
    if (rl >= 2) { div_linear(r, rl, zeta, hw); acc = jac_add(acc, msm_pippenger(hw, s.gg, trimmed_len(hw, rl - 1))); }
    *W = to_affine(acc);
    *random_v = horner(r, rl, zeta);
}

// ------------------------------------------------------------------------------------- round builders
// src/circuit.rs:491-544 with s_id_vals[j] = j (src/prover.rs:205); returns z[0..n) values
static void grand_product_values(const Fr* wflat, const uint64_t* sigma, size_t n, const Fr& beta, const Fr& gamma, Fr* z) {
    std::vector<Fr> num(n), den(n);
    for (size_t i = 0; i < n; ++i) {
        Fr nu = Fr::one(), de = Fr::one();
        for (int k = 0; k < 3; ++k) {
            size_t j = 3 * i + k, sj = sigma[j];
            nu = mul(nu, add(add(wflat[j], mul(beta, from_u64<FrTag>(j))), gamma));
            de = mul(de, add(add(wflat[sj], mul(beta, from_u64<FrTag>(sj))), gamma));
        }
        num[i] = nu; den[i] = de;
    }
    batch_inv(den.data(), n);
    z[0] = Fr::one();
    for (size_t i = 0; i + 1 < n; ++i) z[i + 1] = mul(mul(z[i], num[i]), den[i]);
}
// src/circuit.rs:446-453 + 343-379 summed pointwise; returns 0 on success, 1 if t(X) is not divisible by Z_H(X)
// Public inputs (src/circuit.rs:410-439): constraint[pi_rows[k]] = alpha * (a[pi_rows[k]] - pi_values[k]) in the order of the list (assignment: a
// later entry for the same row replaces an earlier one), interpolated and added to the numerator.  pi_count = 0 is what the prover reaches.
static int quotient(const Fr* a, const Fr* b, const Fr* c, const Fr* qa, const Fr* qm, const Fr* zv, const uint64_t* sigma, int log_n,
                    const Fr& beta, const Fr& gamma, std::vector<Fr>& t_out, const Fr* alpha = nullptr, const uint64_t* pi_rows = nullptr,
                    const Fr* pi_values = nullptr, size_t pi_count = 0) {
    const size_t n = (size_t)1 << log_n;
    std::vector<Fr> gate(n), perm(n), pub(n);
    for (size_t k = 0; k < pi_count; ++k) {
        if (pi_rows[k] >= n) return 2;                                    // the reference indexes out of bounds
        pub[pi_rows[k]] = mul(*alpha, sub(a[pi_rows[k]], pi_values[k]));   // src/circuit.rs:432
    }
    for (size_t i = 0; i < n; ++i) {
        gate[i] = add(mul(qa[i], sub(add(a[i], b[i]), c[i])), mul(qm[i], sub(mul(a[i], b[i]), c[i])));
        const Fr w[3] = {a[i], b[i], c[i]};
        Fr lhs = Fr::one(), rhs = Fr::one();
        for (int k = 0; k < 3; ++k) {
            lhs = mul(lhs, add(add(w[k], mul(beta, from_u64<FrTag>(3 * i + k))), gamma));
            rhs = mul(rhs, add(add(w[k], mul(beta, from_u64<FrTag>(sigma[3 * i + k]))), gamma));
        }
        if (i + 1 < n) perm[i] = sub(mul(zv[i], lhs), mul(zv[i + 1], rhs));
        else perm[i] = sub(mul(mul(zv[i], lhs), inv(rhs)), Fr::one());
    }
    ntt_inplace(gate.data(), log_n, true, nullptr);
    ntt_inplace(perm.data(), log_n, true, nullptr);
    ntt_inplace(pub.data(), log_n, true, nullptr);                          // src/circuit.rs:438 (all zero in the prover, src/prover.rs:224-227)
    for (size_t i = 0; i < n; ++i) gate[i] = add(add(gate[i], perm[i]), pub[i]);   // src/circuit.rs:469
    size_t len = trimmed_len(gate.data(), n);
    // divide_by_vanishing_poly: degree < n, so the quotient is zero and the remainder is the numerator itself
    t_out.clear();
    return len == 0 ? 0 : 1;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------- C ABI for the ctypes wrapper
extern "C" {

int orc_set_threads(int t) { const int old = g_threads; g_threads = t < 1 ? 1 : t; return old; }
void orc_fr_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        Fr x, y, r; memcpy(x.v, a + 4 * i, 32); if (b) memcpy(y.v, b + 4 * i, 32);
        switch (op) { case 0: r = add(x, y); break; case 1: r = sub(x, y); break; case 2: r = mul(x, y); break; case 3: r = inv(x); break; default: r = neg(x); }
        memcpy(out + 4 * i, r.v, 32);
    }
}
void orc_fq_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        Fq x, y, r; memcpy(x.v, a + 4 * i, 32); if (b) memcpy(y.v, b + 4 * i, 32);
        switch (op) { case 0: r = add(x, y); break; case 1: r = sub(x, y); break; case 2: r = mul(x, y); break; case 3: r = inv(x); break; default: r = neg(x); }
        memcpy(out + 4 * i, r.v, 32);
    }
}
void orc_fr_from_canonical(const uint64_t* in, uint64_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) { Fr r = from_canonical<FrTag>(in + 4 * i); memcpy(out + 4 * i, r.v, 32); } }
void orc_fr_to_canonical(const uint64_t* in, uint64_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) { Fr x; memcpy(x.v, in + 4 * i, 32); to_canonical(x, out + 4 * i); } }
void orc_fq_from_canonical(const uint64_t* in, uint64_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) { Fq r = from_canonical<FqTag>(in + 4 * i); memcpy(out + 4 * i, r.v, 32); } }
void orc_fq_to_canonical(const uint64_t* in, uint64_t* out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) { Fq x; memcpy(x.v, in + 4 * i, 32); to_canonical(x, out + 4 * i); } }

void orc_g1_add(const uint64_t* p, int pinf, const uint64_t* q, int qinf, uint64_t* out, int32_t* oinf) {
    Jac a = jac_add_mixed(jac_inf(), load_aff(p, pinf));
    store_aff(to_affine(jac_add_mixed(a, load_aff(q, qinf))), out, oinf);
}
// scalar is a Montgomery-form Fr
void orc_g1_mul(const uint64_t* p, int pinf, const uint64_t* scalar, uint64_t* out, int32_t* oinf) {
    Fr s; memcpy(s.v, scalar, 32); uint64_t k[4]; to_canonical(s, k);
    Aff a = load_aff(p, pinf);
    store_aff(a.inf ? a : to_affine(jac_mul_canonical(a, k)), out, oinf);
}
void orc_g1_generator(uint64_t* out) { Fq x = from_u64<FqTag>(1), y = from_u64<FqTag>(2); memcpy(out, x.v, 32); memcpy(out + 4, y.v, 32); }
void orc_g1_compress(const uint64_t* p, int pinf, uint8_t* out32) { g1_compressed(load_aff(p, pinf), out32); }

// powers_of_g[i] = tau^i * G for i in [first, first+count); incremental fixed-base walk with a 4-bit window table
void orc_srs_powers(const uint64_t* tau, const uint64_t* scale, uint64_t first, uint64_t count, uint64_t* out_xy) {
    Fr t; memcpy(t.v, tau, 32); Fr s; memcpy(s.v, scale, 32);
    Fr cur = s; { uint64_t e[4] = {first, 0, 0, 0}; cur = mul(cur, pow_limbs(t, e)); }
    Aff g; g.x = from_u64<FqTag>(1); g.y = from_u64<FqTag>(2); g.inf = false;
    // table[w][d] = d * 16^w * G
    std::vector<Aff> table(64 * 16);
    Jac base = jac_add_mixed(jac_inf(), g);
    for (int w = 0; w < 64; ++w) {
        Jac acc = jac_inf();
        for (int d = 0; d < 16; ++d) { table[w * 16 + d] = to_affine(acc); acc = jac_add(acc, base); }
        for (int j = 0; j < 4; ++j) base = jac_dbl(base);
    }
    for (uint64_t i = 0; i < count; ++i) {
        uint64_t k[4]; to_canonical(cur, k);
        Jac acc = jac_inf();
        for (int w = 0; w < 64; ++w) { int d = (k[w / 16] >> (4 * (w % 16))) & 15; if (d) acc = jac_add_mixed(acc, table[w * 16 + d]); }
        int32_t inf; store_aff(to_affine(acc), out_xy + 8 * i, &inf);
        cur = mul(cur, t);
    }
}

void orc_ntt(uint64_t* data, int log_n, int inverse, const uint64_t* coset) {
    Fr g; if (coset) memcpy(g.v, coset, 32);
    ntt_inplace((Fr*)data, log_n, inverse != 0, coset ? &g : nullptr);
}
void orc_msm(const uint64_t* scalars, const uint64_t* bases_xy, uint64_t n, uint64_t* out, int32_t* oinf) {
    store_aff(to_affine(msm_pippenger((const Fr*)scalars, bases_xy, n)), out, oinf);
}
void orc_poly_eval(const uint64_t* coeffs, uint64_t len, const uint64_t* x, uint64_t* out) {
    Fr z; memcpy(z.v, x, 32); Fr r = horner((const Fr*)coeffs, len, z); memcpy(out, r.v, 32);
}
void orc_div_linear(const uint64_t* coeffs, uint64_t len, const uint64_t* z, uint64_t* out) {
    Fr zz; memcpy(zz.v, z, 32); if (len >= 2) div_linear((const Fr*)coeffs, len, zz, (Fr*)out);
}
// z coefficients = ifft(z[0..n))
void orc_grand_product(const uint64_t* wflat, const uint64_t* sigma, int log_n, const uint64_t* beta, const uint64_t* gamma, uint64_t* z_vals_out, uint64_t* z_coeffs_out) {
    size_t n = (size_t)1 << log_n; Fr b, g; memcpy(b.v, beta, 32); memcpy(g.v, gamma, 32);
    std::vector<Fr> z(n); grand_product_values((const Fr*)wflat, sigma, n, b, g, z.data());
    if (z_vals_out) memcpy(z_vals_out, z.data(), 32 * n);
    ntt_inplace(z.data(), log_n, true, nullptr);
    memcpy(z_coeffs_out, z.data(), 32 * n);
}
// returns 0 and an all-zero t when divisible; 1 otherwise
int orc_quotient(const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* qa, const uint64_t* qm, const uint64_t* zv, const uint64_t* sigma,
                 int log_n, const uint64_t* beta, const uint64_t* gamma) {
    Fr be, ga; memcpy(be.v, beta, 32); memcpy(ga.v, gamma, 32); std::vector<Fr> t;
    return quotient((const Fr*)a, (const Fr*)b, (const Fr*)c, (const Fr*)qa, (const Fr*)qm, (const Fr*)zv, sigma, log_n, be, ga, t);
}
// the same with public inputs (src/circuit.rs:410-439); returns 2 for a row outside the domain
int orc_quotient_pi(const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* qa, const uint64_t* qm, const uint64_t* zv, const uint64_t* sigma,
                    int log_n, const uint64_t* beta, const uint64_t* gamma, const uint64_t* alpha, const uint64_t* pi_rows, const uint64_t* pi_values, uint64_t pi_count) {
    Fr be, ga, al; memcpy(be.v, beta, 32); memcpy(ga.v, gamma, 32); memcpy(al.v, alpha, 32); std::vector<Fr> t;
    return quotient((const Fr*)a, (const Fr*)b, (const Fr*)c, (const Fr*)qa, (const Fr*)qm, (const Fr*)zv, sigma, log_n, be, ga, t, &al, pi_rows, (const Fr*)pi_values, pi_count);
}
// Quotient of the reference's constraint system on polynomials, as plk_quotient_coset defines it (no counterpart in the reference,
// which never leaves H: src/circuit.rs:442-489).  With a, b, c, q_add, q_mul, z the degree < n interpolants of the column values,
// ID_k of the labels 3i + k and SG_k of sigma(3i + k) (the labels of src/prover.rs:205 and src/circuit.rs:356-368):
//   num(X) = q_add (a + b - c) + q_mul (a b - c) + alpha [ z(X) prod_k (w_k + beta ID_k + gamma) - z(wX) prod_k (w_k + beta SG_k + gamma) ]
//   t(X)   = num(X) / (X^n - 1),  4n coefficients out; returns 1 when the remainder is not zero.
// Checked here WITHOUT a coset: values on the plain domain of size 4n, back to coefficients, schoolbook division by X^n - 1.
int orc_quotient_coset(const uint64_t* a, const uint64_t* b, const uint64_t* c, const uint64_t* qa, const uint64_t* qm, const uint64_t* zv,
                       const uint64_t* sigma, int log_n, const uint64_t* beta, const uint64_t* gamma, const uint64_t* alpha, uint64_t* t_out) {
    const size_t n = (size_t)1 << log_n, n4 = 4 * n;
    Fr be, ga, al; memcpy(be.v, beta, 32); memcpy(ga.v, gamma, 32); memcpy(al.v, alpha, 32);
    auto extend = [&](std::vector<Fr> vals) {   // values on H -> values on the domain of size 4n
        ntt_inplace(vals.data(), log_n, true, nullptr);
        vals.resize(n4, Fr::zero());
        ntt_inplace(vals.data(), log_n + 2, false, nullptr);
        return vals;
    };
    auto column = [&](const uint64_t* src) { std::vector<Fr> v(n); memcpy(v.data(), src, n * 32); return extend(v); };
    std::vector<Fr> A = column(a), B = column(b), Cc = column(c), QA = column(qa), QM = column(qm), Z = column(zv);
    std::vector<Fr> ID[3], SG[3];
    for (int k = 0; k < 3; ++k) {
        std::vector<Fr> id(n), sg(n);
        for (size_t i = 0; i < n; ++i) { id[i] = from_u64<FrTag>(3 * i + k); sg[i] = from_u64<FrTag>(sigma[3 * i + k]); }
        ID[k] = extend(id); SG[k] = extend(sg);
    }
    std::vector<Fr> num(n4);
    for (size_t i = 0; i < n4; ++i) {
        const Fr* w[3] = {&A[i], &B[i], &Cc[i]};
        Fr gate = add(mul(QA[i], sub(add(A[i], B[i]), Cc[i])), mul(QM[i], sub(mul(A[i], B[i]), Cc[i])));
        Fr l = Z[i], r = Z[(i + 4) % n4];                     // z(w x): w = w_{4n}^4
        for (int k = 0; k < 3; ++k) {
            l = mul(l, add(add(*w[k], mul(be, ID[k][i])), ga));
            r = mul(r, add(add(*w[k], mul(be, SG[k][i])), ga));
        }
        num[i] = add(gate, mul(al, sub(l, r)));
    }
    ntt_inplace(num.data(), log_n + 2, true, nullptr);          // degree <= 4n - 4: exact
    std::vector<Fr> q(n4, Fr::zero());
    for (size_t i = n4; i-- > n;) q[i - n] = add(num[i], q[i]);   // f = q (X^n - 1) + r
    int bad = 0;
    for (size_t i = 0; i < n; ++i) if (!add(num[i], q[i]).is_zero()) bad = 1;
    memcpy(t_out, q.data(), n4 * 32);
    return bad;
}

void orc_kzg_commit(const uint64_t* g, uint64_t ng, const uint64_t* gg, const uint64_t* coeffs, uint64_t len, const uint64_t* blind, uint64_t* out, int32_t* oinf) {
    Srs s{g, ng, gg}; store_aff(kzg_commit(s, (const Fr*)coeffs, len, (const Fr*)blind), out, oinf);
}
void orc_kzg_open(const uint64_t* g, uint64_t ng, const uint64_t* gg, const uint64_t* coeffs, uint64_t len, const uint64_t* blind, const uint64_t* zeta,
                  const uint64_t* chal, uint64_t* w_out, int32_t* winf, uint64_t* random_v) {
    Srs s{g, ng, gg}; Fr z, ch, rv; memcpy(z.v, zeta, 32); memcpy(ch.v, chal, 32); Aff W;
    kzg_open(s, (const Fr*)coeffs, len, (const Fr*)blind, z, ch, &W, &rv);
    store_aff(W, w_out, winf); memcpy(random_v, rv.v, 32);
}
void orc_transcript_kat(uint8_t* out32) {
    Transcript t("test protocol"); t.append_message("some label", (const uint8_t*)"some data", 9); t.challenge_bytes("challenge", out32, 32);
}
void orc_empty_beta(uint64_t* out) { Transcript t("plonk_proof"); Fr b = t.challenge_scalar("beta"); memcpy(out, b.v, 32); }

// src/prover.rs:119-336.  polys[7] in the order a,b,c,q_add,q_mul,s_id,s_sigma with lens[7]; blinds = 27 Fr in draw order
// (a,b,c,q_add,q_mul,s_id,s_sigma,z,t); chal = the opening challenge (Montgomery Fr).  proof_out = 1170 bytes:
// 9 evals || 9 commitments (33 B) || 9 openings (65 B) in the field order of src/prover.rs:61-95.
// challenges_out = beta,gamma,alpha,zeta (16 limbs).  stage_s[6] = commit7, grand product(+commit), quotient(+commit), evals, openings, total.
// Returns 0, 1 (t not divisible by Z_H), 2 (wire column shorter than the domain), 3 (SRS too short).
int orc_prove(const uint64_t* g, uint64_t ng, const uint64_t* gg, const uint64_t* const* polys, const uint64_t* lens,
              const uint64_t* public_inputs, uint64_t n_public, const uint64_t* witness_flat, const uint64_t* sigma, int log_n,
              const uint64_t* blinds, const uint64_t* chal, uint8_t* proof_out, uint64_t* challenges_out, double* stage_s) {
    try {
        const size_t n = (size_t)1 << log_n; Srs srs{g, ng, gg};
        const Fr* bl = (const Fr*)blinds; Fr ch; memcpy(ch.v, chal, 32);
        double t0 = now_s(), t_start = t0;
        std::vector<std::vector<Fr>> P(9);
        for (int i = 0; i < 7; ++i) { size_t l = trimmed_len((const Fr*)polys[i], lens[i]); P[i].assign((const Fr*)polys[i], (const Fr*)polys[i] + l); }
        Transcript tr("plonk_proof");
        if (n_public) { std::vector<uint8_t> buf(32 * n_public); for (uint64_t i = 0; i < n_public; ++i) { Fr x; memcpy(x.v, public_inputs + 4 * i, 32); fr_bytes(x, &buf[32 * i]); } tr.append_message("public_inputs", buf.data(), buf.size()); }
        Aff comm[9]; uint8_t cb[33];
        for (int i = 0; i < 7; ++i) comm[i] = kzg_commit(srs, P[i].data(), P[i].size(), bl + 3 * i);
        static const int absorb_order[7] = {3, 4, 0, 1, 2, 5, 6};
        static const char* names[9] = {"a", "b", "c", "q_add", "q_mul", "s_id", "s_sigma", "z", "t"};
        for (int k = 0; k < 7; ++k) { int i = absorb_order[k]; g1_compressed(comm[i], cb); cb[32] = 0; tr.append_message(names[i], cb, 33); }
        Fr beta = tr.challenge_scalar("beta"), gamma = tr.challenge_scalar("gamma");
        double t1 = now_s(); stage_s[0] = t1 - t0; t0 = t1;
        std::vector<Fr> z(n); grand_product_values((const Fr*)witness_flat, sigma, n, beta, gamma, z.data());
        ntt_inplace(z.data(), log_n, true, nullptr);
        P[7].assign(z.begin(), z.begin() + trimmed_len(z.data(), n));
        comm[7] = kzg_commit(srs, P[7].data(), P[7].size(), bl + 21);
        g1_compressed(comm[7], cb); cb[32] = 0; tr.append_message("z", cb, 33);
        Fr alpha = tr.challenge_scalar("alpha");
        t1 = now_s(); stage_s[1] = t1 - t0; t0 = t1;
        if (P[0].size() < n || P[1].size() < n || P[2].size() < n) return 2;
        std::vector<Fr> qa(n, Fr::zero()), qm(n, Fr::zero()), zv(n, Fr::zero());
        std::copy(P[3].begin(), P[3].end(), qa.begin()); std::copy(P[4].begin(), P[4].end(), qm.begin()); std::copy(P[7].begin(), P[7].end(), zv.begin());
        ntt_inplace(qa.data(), log_n, false, nullptr); ntt_inplace(qm.data(), log_n, false, nullptr); ntt_inplace(zv.data(), log_n, false, nullptr);
        if (quotient(P[0].data(), P[1].data(), P[2].data(), qa.data(), qm.data(), zv.data(), sigma, log_n, beta, gamma, P[8])) return 1;
        comm[8] = kzg_commit(srs, P[8].data(), P[8].size(), bl + 24);
        g1_compressed(comm[8], cb); cb[32] = 0; tr.append_message("t", cb, 33);
        Fr zeta = tr.challenge_scalar("zeta");
        t1 = now_s(); stage_s[2] = t1 - t0; t0 = t1;
        Fr ev[9]; uint8_t eb[32];
        for (int i = 0; i < 9; ++i) ev[i] = horner(P[i].data(), P[i].size(), zeta);
        for (int i = 0; i < 9; ++i) { char label[24] = "eval_"; strcat(label, names[i]); fr_bytes(ev[i], eb); tr.append_message(label, eb, 32); }
        t1 = now_s(); stage_s[3] = t1 - t0; t0 = t1;
        Aff W[9]; Fr rv[9];
        for (int i = 0; i < 9; ++i) kzg_open(srs, P[i].data(), P[i].size(), bl + 3 * i, zeta, ch, &W[i], &rv[i]);
        t1 = now_s(); stage_s[4] = t1 - t0; stage_s[5] = t1 - t_start;
        uint8_t* o = proof_out;
        for (int i = 0; i < 9; ++i) { fr_bytes(ev[i], o); o += 32; }
        for (int i = 0; i < 9; ++i) { g1_compressed(comm[i], o); o[32] = 0; o += 33; }
        for (int i = 0; i < 9; ++i) { g1_compressed(W[i], o); o[32] = 1; fr_bytes(rv[i], o + 33); o += 65; }
        memcpy(challenges_out, beta.v, 32); memcpy(challenges_out + 4, gamma.v, 32); memcpy(challenges_out + 8, alpha.v, 32); memcpy(challenges_out + 12, zeta.v, 32);
        return 0;
    } catch (const std::exception&) { return 3; }
}

}  // extern "C"
