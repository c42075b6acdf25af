// Radix-2 NTT over BN254 Fr for gfx950: natural order in and out (ark-poly fft / ifft semantics).
//
// Stockham passes of radix R = 2^log_r <= 1024.  A workgroup stages T independent R-point transforms
// (T * R = 4096 elements, 147 KB of LDS as nine-limb values) and runs all log_r butterfly stages out of LDS,
// so the data crosses HBM once per pass: two passes up to 2^20, three up to 2^30.  Loads and stores move
// whole 32-byte elements with lanes on consecutive elements (T consecutive elements per row, or a whole row
// of R on the first pass), so every wavefront access covers full 128-byte lines.
// A pass reads   in[j + t * N/R] * w_{Ns R}^{(j mod Ns) t}   for t < R   (Ns = product of the earlier radices)
// and writes     out[(j div Ns) Ns R + (j mod Ns) + u Ns]    for u < R.
// Butterflies are decimation in time on bit-reversed LDS positions: the product with the twiddle comes first,
// so lazily reduced values grow by 2p per stage only (bound 22 after ten stages).
#include "common.cuh"
#include "host_math.hpp"
#include "kernels.h"
#include <mutex>

namespace plk {

struct NttScale {
    uint32_t w[8];
};
constexpr int NTT_TILE = 4096;
constexpr int NTT_THREADS = 1024;

__device__ __forceinline__ fe lds_get(const uint32_t* lds, uint32_t idx) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = lds[idx * 9 + i];
    return r;
}
__device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t idx, const fe& a) {
#pragma unroll
    for (int i = 0; i < 9; ++i) lds[idx * 9 + i] = a.v[i];
}
// x * w^e (inverse: w^-e) with the half-size table w^i, i < n/2, and w^(n/2) = -1.  Result normalized, bound 2.
__device__ __forceinline__ fe mul_tw(const fe& x, const uint32_t* __restrict__ W, uint32_t e, uint32_t log_n, int inverse) {
    const uint32_t n = 1u << log_n, half = n >> 1;
    if (inverse) e = (n - e) & (n - 1);
    const bool neg = e >= half;
    if (neg) e -= half;
    fe y = mul<FrP>(x, ld_fe(W + 8ull * e));
    if (neg) { y = sub2<FrP>(fe_zero(), y); norm(y); }
    return y;
}

__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ W,
                                                           uint32_t log_n, uint32_t log_r, uint32_t log_ns, uint32_t log_t, int inverse, int last,
                                                           int use_scale, NttScale scale /* x*2^256 words */) {
    __shared__ uint32_t lds[NTT_TILE * 9];
    const uint32_t T = 1u << log_t, R = 1u << log_r, ns_mask = (1u << log_ns) - 1;
    const uint32_t stride = 1u << (log_n - log_r), tile = T << log_r, wg = blockIdx.x;
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        const uint32_t jj = e & (T - 1), t = e >> log_t, j = wg * T + jj;
        fe x = ld_fe(in + 8ull * (j + (uint64_t)t * stride));
        const uint32_t k = j & ns_mask;
        if (t > 0 && k > 0) x = mul_tw(x, W, (k * t) << (log_n - log_ns - log_r), log_n, inverse);
        const uint32_t pos = log_r ? (__brev(t) >> (32 - log_r)) : 0;
        lds_put(lds, pos * T + jj, x);
    }
    __syncthreads();
    for (uint32_t s = 1; s <= log_r; ++s) {
        const uint32_t half = 1u << (s - 1);
        for (uint32_t bf = threadIdx.x; bf < (tile >> 1); bf += NTT_THREADS) {
            const uint32_t jj = bf & (T - 1), pi = bf >> log_t;
            const uint32_t i = pi & (half - 1), p0 = ((pi >> (s - 1)) << s) + i, p1 = p0 + half;
            const fe u = lds_get(lds, p0 * T + jj);
            fe v = lds_get(lds, p1 * T + jj);
            if (s > 1) v = mul_tw(v, W, i << (log_n - s), log_n, inverse);
            fe r0 = add(u, v), r1 = sub2<FrP>(u, v);
            norm(r0);
            norm(r1);
            lds_put(lds, p0 * T + jj, r0);
            lds_put(lds, p1 * T + jj, r1);
        }
        __syncthreads();
    }
    const fe sc = use_scale ? mul<FrP>(unpack(scale.w), fe_const(FrP::TO_RP)) : fe_const(FrP::ONE);
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t jj, u;
        if (log_ns == 0) { u = e & (R - 1); jj = e >> log_r; } else { jj = e & (T - 1); u = e >> log_t; }
        const uint32_t j = wg * T + jj;
        const uint64_t j0 = ((uint64_t)(j >> log_ns) << (log_ns + log_r)) | (j & ns_mask);
        const fe y = mul<FrP>(lds_get(lds, u * T + jj), sc);
        uint32_t* dst = out + 8ull * (j0 + ((uint64_t)u << log_ns));
        if (last) st_fe<FrP>(dst, y);
        else { uint32_t w[8]; pack(y, w); store8(dst, w); }
    }
}

// ------------------------------------------------------------------------------------------ sizes >= 2^11
// Same Stockham passes, different inner structure.  A workgroup of 256 threads owns T * R = 2048 elements (72 KB of LDS, two
// workgroups per CU so that one loads or stores while the other computes).  Every thread keeps eight elements in registers and
// runs up to three butterfly stages on them between two visits to LDS: the first three stages work on the values as they arrive
// from HBM, the last ones hand their results straight to the store.  The multiplication that brings a lazily reduced value back
// below 2p before it is stored carries the factor the NEXT pass would apply on load (w_{Ns' R'}^{(j mod Ns') t}), or n^-1 in
// the final pass of an inverse transform, so loads never multiply.  Twiddles of the butterfly stages come from a per-pass table
// w_R^e (e < R/2) in LDS.  Stages whose twiddle is 1 (all of stage 1, half of stage 2, a quarter of stage 3) skip the product.
//
// Bounds (fp29.cuh vocabulary).  Loaded values: normalized, bound 2.  First three stages: at most bound 16, limbs < 3.5 * 2^30.
// Every later stage adds 2 to the bound and 2^30 to the limbs; values are normalized whenever they go to LDS, so a product never
// sees an operand limb above 2.5 * 2^30 < 2^31.6, and the bound stays below 16 + 2 * 6 = 28 (times 2 for the twiddle: 56 <= 169).
constexpr int NTT8_TILE = 2048;
constexpr int NTT8_THREADS = 256;
constexpr int NTT8_MAX_LOG_R = 8;   // 2048 * 36 B of data + 128 * 36 B of twiddles: two workgroups fit the 160 KB of a CU
constexpr int NTT8_LDS_BYTES = (NTT8_TILE * 9 + (1 << (NTT8_MAX_LOG_R - 1)) * 9) * 4;

struct Ntt8Pass {
    uint32_t log_n, log_r, log_ns, log_t;
    uint32_t nsub, q[4];        // butterfly stages per visit of the registers; q[0] == 3, sum == log_r
    uint32_t next_log_r;        // radix of the following pass, 0 in the last pass
    int32_t inverse, use_scale;
};

// w^e (inverse: w^-e) as a multiplication operand: bound 2, limbs < 2^31 (2p - w is left unnormalized)
__device__ __forceinline__ fe twiddle_operand(const uint32_t* __restrict__ W, uint32_t e, uint32_t log_n, int inverse) {
    const uint32_t n = 1u << log_n, half = n >> 1;
    if (inverse) e = (n - e) & (n - 1);
    const bool neg = e >= half;
    if (neg) e -= half;
    fe w = ld_fe(W + 8ull * e);
    if (neg) w = sub2<FrP>(fe_zero(), w);
    return w;
}

// stages 1..3 on positions 0..7 of a group whose twiddles are the constants 1, w_4, w_8, w_8^2, w_8^3
__device__ __forceinline__ void ntt8_first_stages(fe* x, const uint32_t* tw, uint32_t log_r) {
#pragma unroll
    for (int a = 0; a < 8; a += 2) {                      // stage 1: twiddle 1
        const fe u = x[a], v = x[a + 1];
        x[a] = add(u, v);
        x[a + 1] = sub2<FrP>(u, v);                       // v normalized, bound 2
    }
#pragma unroll
    for (int a = 0; a < 8; a += 4) {                      // stage 2
        {
            const fe u = x[a];
            fe v = x[a + 2];
            norm(v);                                      // bound 4
            x[a] = add(u, v);
            x[a + 2] = sub4<FrP>(u, v);
        }
        {
            const fe u = x[a + 1];
            const fe t = mul<FrP>(x[a + 3], lds_get(tw, 1u << (log_r - 2)));
            x[a + 1] = add(u, t);
            x[a + 3] = sub2<FrP>(u, t);
        }
    }
    {                                                     // stage 3
        const fe u = x[0];
        fe v = x[4];
        norm(v);                                          // bound 8
        x[0] = add(u, v);
        x[4] = sub8<FrP>(u, v);
    }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        const fe u = x[i];
        const fe t = mul<FrP>(x[i + 4], lds_get(tw, (uint32_t)i << (log_r - 3)));
        x[i] = add(u, t);
        x[i + 4] = sub2<FrP>(u, t);
    }
}

// stage s+D+1 of a group of 2^Q elements x[0..2^Q) at positions base + m 2^s; base_low = base mod 2^s
template <int Q, int D>
__device__ __forceinline__ void ntt8_stage(fe* x, const uint32_t* tw, uint32_t base_low, uint32_t s, uint32_t log_r) {
#pragma unroll
    for (int low = 0; low < (1 << D); ++low) {
        const uint32_t i = base_low + ((uint32_t)low << s);
        const fe w = lds_get(tw, i << (log_r - s - D - 1));
#pragma unroll
        for (int hi = 0; hi < (1 << (Q - D - 1)); ++hi) {
            const int m0 = (hi << (D + 1)) | low, m1 = m0 | (1 << D);
            const fe u = x[m0];
            const fe t = mul<FrP>(x[m1], w);
            x[m0] = add(u, t);
            x[m1] = sub2<FrP>(u, t);
        }
    }
}
template <int Q>
__device__ __forceinline__ void ntt8_stages(fe* x, const uint32_t* tw, uint32_t base_low, uint32_t s, uint32_t log_r) {
    ntt8_stage<Q, 0>(x, tw, base_low, s, log_r);
    if constexpr (Q > 1) ntt8_stage<Q, 1>(x, tw, base_low, s, log_r);
    if constexpr (Q > 2) ntt8_stage<Q, 2>(x, tw, base_low, s, log_r);
}

// one visit of the registers after the first: every thread takes 8 >> Q groups of 2^Q elements, one after the other
template <int Q, bool OUT_F, bool TW_T>
__device__ __forceinline__ void ntt8_subpass(uint32_t* lds, const uint32_t* tw, const Ntt8Pass& a, uint32_t s, bool final_sub, const uint32_t* __restrict__ W,
                                             uint32_t* __restrict__ out, const fe& sc, const uint32_t* __restrict__ out_factors) {
    constexpr int G = 8 >> Q, E = 1 << Q;
    const uint32_t log_t = a.log_t, T = 1u << log_t, log_r = a.log_r, ns_mask = (1u << a.log_ns) - 1;
#pragma unroll 1
    for (int c = 0; c < G; ++c) {
        const uint32_t gid = threadIdx.x + c * NTT8_THREADS;
        uint32_t jj, gpos;
        if (final_sub && a.log_ns == 0) { gpos = gid & ((1u << (log_r - Q)) - 1); jj = gid >> (log_r - Q); }   // stores run along u
        else { jj = gid & (T - 1); gpos = gid >> log_t; }                                                      // stores run along j
        const uint32_t base_low = gpos & ((1u << s) - 1);
        const uint32_t pos0 = ((gpos >> s) << (s + Q)) | base_low;
        fe x[E];
#pragma unroll
        for (int m = 0; m < E; ++m) x[m] = lds_get(lds, ((pos0 + ((uint32_t)m << s)) << log_t) + jj);
        ntt8_stages<Q>(x, tw, base_low, s, log_r);
#pragma unroll
        for (int m = 0; m < E; ++m) norm(x[m]);
        if (!final_sub) {
#pragma unroll
            for (int m = 0; m < E; ++m) lds_put(lds, ((pos0 + ((uint32_t)m << s)) << log_t) + jj, x[m]);
            continue;
        }
        const uint64_t j = (uint64_t)blockIdx.x * T + jj;
        const uint64_t j0 = ((j >> a.log_ns) << (a.log_ns + log_r)) | (j & ns_mask);
        static_for<E>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const uint32_t u = pos0 + ((uint32_t)m << s);
            const uint64_t o = j0 + ((uint64_t)u << a.log_ns);
            uint32_t* dst = out + 8ull * o;
            if (a.next_log_r == 0) {
                if constexpr (OUT_F) st_fe<FrP>(dst, mul<FrP>(x[m], ld_fe(out_factors + 8ull * o)));   // inverse coset: n^-1 g^-o
                else st_fe<FrP>(dst, mul<FrP>(x[m], sc));
            } else {
                fe y;
                if constexpr (TW_T) {
                    y = mul<FrP>(x[m], ld_fe(out_factors + 8ull * o));   // the next pass's factors, laid out in this pass's output order
                } else {
                    const uint32_t lns2 = a.log_ns + log_r;
                    const uint32_t t2 = (uint32_t)(o >> (a.log_n - a.next_log_r)), k2 = (uint32_t)o & ((1u << lns2) - 1);
                    fe w = sc;   // 1
                    if (t2 != 0 && k2 != 0) w = twiddle_operand(W, (k2 * t2) << (a.log_n - lns2 - a.next_log_r), a.log_n, a.inverse);
                    y = mul<FrP>(x[m], w);
                }
                uint32_t wd[8];
                pack(y, wd);
                store8(dst, wd);
            }
        });
    }
}

// IN_F: multiply by in_factors[i] while loading (first pass of a forward coset transform); OUT_F: the final product takes
// out_factors[o] instead of the scale (last pass of an inverse coset transform); TW_T: a pass that is not the last takes the
// next pass's twiddles from out_factors[o], a table in output order that is streamed with the data, instead of gathering
// them from the n/2 table of powers (32-byte reads all over a table of up to n * 16 bytes: twice the data's own traffic).
template <bool IN_F, bool OUT_F, bool TW_T>
__global__ void __launch_bounds__(NTT8_THREADS) k_ntt_pass8(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, const uint32_t* __restrict__ W, Ntt8Pass a,
                                                            NttScale scale /* x*2^256 words */, const uint32_t* __restrict__ in_factors,
                                                            const uint32_t* __restrict__ out_factors) {
    // LDS handed over at launch: told the size, the compiler sizes the register allocation for the two workgroups per CU that
    // 78 KB permit (169 registers allocated for 157 used), and the kernel's waves then do not fit into the 160 registers that two
    // accumulation waves leave free on a SIMD.
    extern __shared__ uint32_t ntt8_lds[];
    uint32_t* lds = ntt8_lds;
    uint32_t* tw = ntt8_lds + NTT8_TILE * 9;
    const uint32_t log_r = a.log_r, log_t = a.log_t, T = 1u << log_t;
    for (uint32_t e = threadIdx.x; e < (1u << (log_r - 1)); e += NTT8_THREADS) {
        fe w = twiddle_operand(W, e << (a.log_n - log_r), a.log_n, a.inverse);
        norm(w);
        lds_put(tw, e, w);
    }
    {
        fe x[8];
        const uint32_t jj = threadIdx.x & (T - 1), g = threadIdx.x >> log_t;   // g < R / 8: low bits of the row index
        const uint64_t j = (uint64_t)blockIdx.x * T + jj;
        const uint32_t log_stride = a.log_n - log_r;
        // position m of the group holds row brev3(m) R/8 + g.  Written out eight times: with the coset product inside, the compiler
        // keeps a loop here and moves x[] to scratch memory.
#define NTT8_LOAD(m)                                                                                                          \
        {                                                                                                                     \
            const uint32_t t = ((uint32_t)((((m) & 1) << 2) | ((m) & 2) | ((m) >> 2)) << (log_r - 3)) | g;                     \
            const uint64_t at = j + ((uint64_t)t << log_stride);                                                              \
            if constexpr (IN_F) x[m] = mul<FrP>(ld_fe(in + 8ull * at), ld_fe(in_factors + 8ull * at)); /* coset: x_i g^i */   \
            else x[m] = ld_fe(in + 8ull * at);                                                                                \
        }
        NTT8_LOAD(0) NTT8_LOAD(1) NTT8_LOAD(2) NTT8_LOAD(3) NTT8_LOAD(4) NTT8_LOAD(5) NTT8_LOAD(6) NTT8_LOAD(7)
#undef NTT8_LOAD
        __syncthreads();   // twiddle table complete
        ntt8_first_stages(x, tw, log_r);
        const uint32_t pos_hi = log_r > 3 ? (__brev(g) >> (32 - (log_r - 3))) : 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            norm(x[m]);
            lds_put(lds, (((pos_hi << 3) | (uint32_t)m) << log_t) + jj, x[m]);
        }
    }
    __syncthreads();
    const fe sc = a.use_scale ? mul<FrP>(unpack(scale.w), fe_const(FrP::TO_RP)) : fe_const(FrP::ONE);
    uint32_t s = 3;
    for (uint32_t sp = 1; sp < a.nsub; ++sp) {
        const bool final_sub = sp + 1 == a.nsub;
        const uint32_t q = a.q[sp];
        if (q == 3) ntt8_subpass<3, OUT_F, TW_T>(lds, tw, a, s, final_sub, W, out, sc, out_factors);
        else if (q == 2) ntt8_subpass<2, OUT_F, TW_T>(lds, tw, a, s, final_sub, W, out, sc, out_factors);
        else ntt8_subpass<1, OUT_F, TW_T>(lds, tw, a, s, final_sub, W, out, sc, out_factors);
        s += q;
        if (!final_sub) __syncthreads();
    }
}

// W[i] = w^i for i < n/2 from pow2[k] = w^(2^k) (x*2^256 words)
__global__ void __launch_bounds__(256) k_ntt_twiddles(uint32_t* __restrict__ W, uint32_t half, const uint32_t* __restrict__ pow2) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= half) return;
    const fe to_rp = fe_const(FrP::TO_RP);
    fe w = fe_const(FrP::ONE);
    for (int k = 0; k < 30; ++k)
        if ((i >> k) & 1) w = mul<FrP>(w, mul<FrP>(ld_fe(pow2 + 8 * k), to_rp));
    st_fe<FrP>(W + 8ull * i, w);
}
// g^(2^k) for k < 40 and a base factor, x*2^256 words; passed by value so that no staging buffer or wait is needed
struct PowerArgs {
    uint32_t pow2[40][8];
    uint32_t base[8];
};
__device__ __forceinline__ fe power_at(const PowerArgs& a, uint64_t i0) {
    const fe to_rp = fe_const(FrP::TO_RP);
    fe cur = mul<FrP>(unpack(a.base), to_rp);
    for (int k = 0; k < 40; ++k)
        if ((i0 >> k) & 1) cur = mul<FrP>(cur, mul<FrP>(unpack(a.pow2[k]), to_rp));
    return cur;   // base * g^i0 in the 2^261 form
}
// data[i] *= base * g^i.  Each thread walks 8 consecutive elements.
__global__ void __launch_bounds__(256) k_scale_powers(uint32_t* __restrict__ data, uint64_t n, PowerArgs a) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    fe cur = power_at(a, i0);
    const fe g = mul<FrP>(unpack(a.pow2[0]), fe_const(FrP::TO_RP));
    for (int k = 0; k < 8 && i0 + k < n; ++k) {
        st_fe<FrP>(data + 8 * (i0 + k), mul<FrP>(ld_fe(data + 8 * (i0 + k)), cur));
        cur = mul<FrP>(cur, g);
    }
}
// out[i] = base * g^i in the 2^261 form (a multiplication operand that keeps the other factor's form)
__global__ void __launch_bounds__(256) k_power_table(uint32_t* __restrict__ out, uint64_t n, PowerArgs a) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    fe cur = power_at(a, i0);
    const fe g = mul<FrP>(unpack(a.pow2[0]), fe_const(FrP::TO_RP));
    for (int k = 0; k < 8 && i0 + k < n; ++k) {
        st_fe<FrP>(out + 8 * (i0 + k), cur);
        cur = mul<FrP>(cur, g);
    }
}
static PowerArgs power_args(const host::el& g, const host::el& base) {
    PowerArgs a;
    host::el p = g;
    for (int k = 0; k < 40; ++k) { memcpy(a.pow2[k], p.v, 32); p = host::mul(host::FR, p, p); }
    memcpy(a.base, base.v, 32);
    return a;
}

// G-point transforms down the columns of a [G][cols] array, G = 2^LG <= 8, in place: row k1 receives
// sum_g w_G^(+-g k1) row g.  One thread per column, all G values in registers.  No scaling in either direction.
// Values are canonical on the way in and out.
template <int LG, int D>
__device__ __forceinline__ void cross_stage(fe* x, const uint32_t* __restrict__ W, int inverse) {
#pragma unroll
    for (int low = 0; low < (1 << D); ++low) {
        const fe w = twiddle_operand(W, (uint32_t)low << (LG - D - 1), LG, inverse);
#pragma unroll
        for (int hi = 0; hi < (1 << (LG - D - 1)); ++hi) {
            const int m0 = (hi << (D + 1)) | low, m1 = m0 | (1 << D);
            const fe u = x[m0];
            const fe t = mul<FrP>(x[m1], w);
            x[m0] = add(u, t);
            x[m1] = sub2<FrP>(u, t);
        }
    }
}
template <int LG>
__global__ void __launch_bounds__(256) k_ntt_cross(uint32_t* __restrict__ data, uint64_t cols, const uint32_t* __restrict__ W /* w_G^i, i < G/2 */, int inverse) {
    constexpr int G = 1 << LG;
    const uint64_t col = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    fe x[G];
#pragma unroll
    for (int m = 0; m < G; ++m) {
        int row = 0;
#pragma unroll
        for (int b = 0; b < LG; ++b) row |= ((m >> b) & 1) << (LG - 1 - b);
        x[m] = ld_fe(data + 8ull * ((uint64_t)row * cols + col));
    }
    cross_stage<LG, 0>(x, W, inverse);
    if constexpr (LG > 1) cross_stage<LG, 1>(x, W, inverse);
    if constexpr (LG > 2) {   // limbs grow by 2^30 per stage: bring them back below 2^29 after two stages
#pragma unroll
        for (int m = 0; m < G; ++m) norm(x[m]);
        cross_stage<LG, 2>(x, W, inverse);
    }
    const fe one = fe_const(FrP::ONE);
    static_for<G>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        norm(x[m]);
        st_fe<FrP>(data + 8ull * ((uint64_t)m * cols + col), mul<FrP>(x[m], one));
    });
}

// out[o] = the factor the pass after this one applies to element o: w^(+-(o mod Ns') (o div N/R') N/(Ns' R')), canonical, x*2^261 form
__global__ void __launch_bounds__(256) k_pass_twiddles(uint32_t* __restrict__ out, const uint32_t* __restrict__ W, uint32_t log_n, uint32_t next_log_r, uint32_t lns2,
                                                       int inverse) {
    const uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (1ull << log_n)) return;
    const uint32_t t2 = (uint32_t)(o >> (log_n - next_log_r)), k2 = (uint32_t)o & ((1u << lns2) - 1);
    fe w = fe_const(FrP::ONE);
    if (t2 != 0 && k2 != 0) { w = twiddle_operand(W, (k2 * t2) << (log_n - lns2 - next_log_r), log_n, inverse); norm(w); }
    st_fe<FrP>(out + 8ull * o, w);
}

static host::el root_of_unity(uint32_t log_n) {
    static const uint64_t ROOT28[4] = {0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL, 0x1860ef942963f9e7ULL};
    host::el w = host::el_load(ROOT28);
    for (uint32_t i = log_n; i < 28; ++i) w = host::mul(host::FR, w, w);
    return w;
}

int32_t ntt_twiddles(plk_ctx* ctx, uint32_t log_n, const uint32_t** out) {
    auto it = ctx->twiddles.find((int)log_n);
    if (it != ctx->twiddles.end()) { *out = it->second; return PLK_OK; }
    const uint32_t half = 1u << (log_n - 1);
    uint64_t p2[30 * 4];
    host::el w = root_of_unity(log_n);
    for (int k = 0; k < 30; ++k) { memcpy(p2 + 4 * k, w.v, 32); w = host::mul(host::FR, w, w); }
    uint32_t *p2_dev = nullptr, *W = nullptr;
    PLK_HIP(hipMalloc(&p2_dev, sizeof(p2)));
    PLK_HIP(hipMalloc(&W, (size_t)half * 32));
    PLK_HIP(hipMemcpy(p2_dev, p2, sizeof(p2), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ntt_twiddles, dim3((half + 255) / 256), dim3(256), 0, ctx->stream, W, half, p2_dev);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    PLK_HIP(hipFree(p2_dev));
    ctx->twiddles[(int)log_n] = W;
    *out = W;
    return PLK_OK;
}
void ntt_release(plk_ctx* ctx) {
    for (auto& kv : ctx->twiddles) hipFree(kv.second);
    ctx->twiddles.clear();
    for (auto& kv : ctx->coset_factors) hipFree(kv.second);
    ctx->coset_factors.clear();
    for (int k = 0; k < 2; ++k) {
        hipFree(ctx->ntt_tmp[k][0]); hipFree(ctx->ntt_tmp[k][1]);
        ctx->ntt_tmp[k][0] = ctx->ntt_tmp[k][1] = nullptr; ctx->ntt_tmp_cap[k] = 0;
    }
}

static int32_t scale_powers(plk_ctx* ctx, hipStream_t st, uint32_t* data, uint64_t n, const host::el& g, const host::el& base) {
    hipLaunchKernelGGL(k_scale_powers, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, st, data, n, power_args(g, base));
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// Per-element factors of a coset transform of 2^log_n >= 2^11 elements, built once per (size, direction, generator) and kept:
// forward g^i (applied while the first pass loads), inverse n^-1 g^-i (applied while the last pass stores).
static int32_t coset_factors(plk_ctx* ctx, hipStream_t st, uint32_t log_n, bool inverse, const uint64_t* gen, const uint32_t** out) {
    std::vector<uint64_t> key = {log_n, inverse ? 1ull : 0ull, gen[0], gen[1], gen[2], gen[3]};
    auto it = ctx->coset_factors.find(key);
    if (it != ctx->coset_factors.end()) { *out = it->second; return PLK_OK; }
    // The tables are keyed by the generator, which the caller chooses: a caller that keeps changing it must not fill the device.
    // At most eight are kept; past that all of them go (after everything that may still read one has run) and are rebuilt on demand.
    constexpr size_t MAX_GENERATOR_TABLES = 8;
    size_t held = 0;
    for (const auto& kv : ctx->coset_factors) held += kv.first[0] != ~0ull ? 1 : 0;
    if (held >= MAX_GENERATOR_TABLES) {
        PLK_HIP(hipDeviceSynchronize());
        for (auto e = ctx->coset_factors.begin(); e != ctx->coset_factors.end();) {
            if (e->first[0] != ~0ull) { (void)hipFree(e->second); e = ctx->coset_factors.erase(e); }
            else ++e;
        }
    }
    const uint64_t n = 1ull << log_n;
    host::el g = host::el_load(gen), base = host::one(host::FR);
    if (inverse) {
        if (host::is_zero(g)) { ctx->err = "coset generator is zero"; return PLK_ERR_INVERSE_OF_ZERO; }
        g = host::inv(host::FR, g);
        base = host::inv(host::FR, host::from_u64(host::FR, n));
    }
    uint32_t* table = nullptr;
    if (hipMalloc(&table, n * 32) != hipSuccess) { ctx->err = "out of device memory for the coset factors"; return PLK_ERR_ALLOC; }
    hipLaunchKernelGGL(k_power_table, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, table, n, power_args(g, base));
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(st));   // built once, then read by transforms on either compute stream
    ctx->coset_factors[key] = table;
    *out = table;
    return PLK_OK;
}

// Twiddles between pass p and pass p + 1 of a transform of 2^log_n elements in pass p's output order, built once and kept
// (n * 32 bytes per pass boundary and direction; HBM capacity buys the coalesced access).
constexpr uint32_t NTT_TABLE_MAX_LOG_N = 25;
static int32_t pass_twiddles(plk_ctx* ctx, hipStream_t st, const uint32_t* W, uint32_t log_n, bool inverse, uint32_t p, uint32_t next_log_r, uint32_t lns2,
                             const uint32_t** out) {
    std::vector<uint64_t> key = {0xffffffffffffffffull, log_n, inverse ? 1ull : 0ull, p, next_log_r, lns2};
    auto it = ctx->coset_factors.find(key);
    if (it != ctx->coset_factors.end()) { *out = it->second; return PLK_OK; }
    const uint64_t n = 1ull << log_n;
    uint32_t* table = nullptr;
    if (hipMalloc(&table, n * 32) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return PLK_OK; }   // no room: the pass gathers instead
    hipLaunchKernelGGL(k_pass_twiddles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, table, W, log_n, next_log_r, lns2, inverse ? 1 : 0);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(st));   // built once, then read by transforms on either compute stream
    ctx->coset_factors[key] = table;
    *out = table;
    return PLK_OK;
}

int32_t ntt_run(plk_ctx* ctx, hipStream_t st, uint32_t* data, uint32_t log_n, bool inverse, const uint64_t* coset_gen) {
    if (log_n > 27) { ctx->err = "NTT size above 2^27"; return PLK_ERR_INVALID_ARG; }
    if (coset_gen && inverse && host::is_zero(host::el_load(coset_gen))) { ctx->err = "coset generator is zero"; return PLK_ERR_INVERSE_OF_ZERO; }
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) return PLK_OK;
    const uint32_t* W;
    PLK_TRY(ntt_twiddles(ctx, log_n, &W));
    // pass plan: radices as even as possible, at most 2^10 per pass below 2^11 elements (k_ntt_pass), 2^8 from there on (k_ntt_pass8)
    const bool big = log_n >= 11;
    uint32_t lr[4], lns[4], lt[4], npass;
    if (log_n <= 10) { npass = 1; lr[0] = log_n; lns[0] = 0; lt[0] = 0; }
    else {
        npass = (log_n + NTT8_MAX_LOG_R - 1) / NTT8_MAX_LOG_R;
        const uint32_t base = log_n / npass, rem = log_n % npass;
        uint32_t acc = 0;
        for (uint32_t p = 0; p < npass; ++p) {
            lr[p] = base + (p < rem ? 1 : 0);
            lns[p] = acc;
            lt[p] = 11 - lr[p];
            acc += lr[p];
        }
    }
    const int set = st == ctx->st_red ? 1 : 0;   // detached transforms have buffers of their own
    uint32_t** tmp = ctx->ntt_tmp[set];
    if (npass > 1 && ctx->ntt_tmp_cap[set] < n) {
        PLK_HIP(hipStreamSynchronize(st));
        hipFree(tmp[0]); hipFree(tmp[1]);
        tmp[0] = tmp[1] = nullptr; ctx->ntt_tmp_cap[set] = 0;
        PLK_HIP(hipMalloc(&tmp[0], n * 32));
        PLK_HIP(hipMalloc(&tmp[1], n * 32));
        ctx->ntt_tmp_cap[set] = n;
    }
    const uint32_t* factors = nullptr;
    if (coset_gen && big) PLK_TRY(coset_factors(ctx, st, log_n, inverse, coset_gen, &factors));
    if (coset_gen && !big && !inverse) PLK_TRY(scale_powers(ctx, st, data, n, host::el_load(coset_gen), host::one(host::FR)));
    NttScale ninv;
    memset(&ninv, 0, sizeof(ninv));
    if (inverse) {
        const host::el v = host::inv(host::FR, host::from_u64(host::FR, n));
        memcpy(ninv.w, v.v, 32);
    }
    prof_begin(ctx, st, "ntt");
    const uint32_t* src = data;
    for (uint32_t p = 0; p < npass; ++p) {
        const bool last = p + 1 == npass;
        uint32_t* dst = last ? data : tmp[p & 1];
        if (!big) {
            hipLaunchKernelGGL(k_ntt_pass, dim3(1), dim3(NTT_THREADS), 0, st, src, dst, W, log_n, lr[p], lns[p], lt[p], inverse ? 1 : 0, last ? 1 : 0,
                               (last && inverse) ? 1 : 0, ninv);
        } else {
            Ntt8Pass a;
            a.log_n = log_n; a.log_r = lr[p]; a.log_ns = lns[p]; a.log_t = lt[p];
            a.next_log_r = last ? 0 : lr[p + 1];
            a.inverse = inverse ? 1 : 0; a.use_scale = (last && inverse) ? 1 : 0;
            // three stages per visit of the registers, the remainder as 2 or 2 + 2
            uint32_t left = lr[p] - 3;
            a.nsub = 1; a.q[0] = 3; a.q[1] = a.q[2] = a.q[3] = 0;
            while (left > 0) {
                const uint32_t q = (left == 4 || left == 2) ? 2 : (left >= 3 ? 3 : 1);
                a.q[a.nsub++] = q;
                left -= q;
            }
            const dim3 grid((uint32_t)(n / NTT8_TILE)), block(NTT8_THREADS);
            // more than the 64 KB a kernel may take without asking: asked for once per device (function attributes belong to the device
            // that is current when they are set)
            static std::mutex lds_mutex;
            static bool lds_asked[64] = {}, lds_granted[64] = {};
            bool lds_allowed;
            {
                std::lock_guard<std::mutex> lock(lds_mutex);
                const int dev = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
                if (!lds_asked[dev]) {
                    bool ok = true;
                    auto allow = [&](const void* f) { ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, NTT8_LDS_BYTES) == hipSuccess; };
                    allow(reinterpret_cast<const void*>(k_ntt_pass8<false, true, false>));
                    allow(reinterpret_cast<const void*>(k_ntt_pass8<false, false, false>));
                    allow(reinterpret_cast<const void*>(k_ntt_pass8<true, false, true>));
                    allow(reinterpret_cast<const void*>(k_ntt_pass8<false, false, true>));
                    allow(reinterpret_cast<const void*>(k_ntt_pass8<true, false, false>));
                    lds_asked[dev] = true; lds_granted[dev] = ok;
                }
                lds_allowed = lds_granted[dev];
            }
            if (!lds_allowed) { ctx->err = "k_ntt_pass8: 78 KB of LDS refused"; return PLK_ERR_HIP; }
            const uint32_t* tw_next = nullptr;
            if (!last && log_n <= NTT_TABLE_MAX_LOG_N) PLK_TRY(pass_twiddles(ctx, st, W, log_n, inverse, p, lr[p + 1], lns[p] + lr[p], &tw_next));
            const bool in_f = factors && p == 0 && !inverse;
            if (last) {
                if (factors && inverse) hipLaunchKernelGGL((k_ntt_pass8<false, true, false>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, nullptr, factors);
                else hipLaunchKernelGGL((k_ntt_pass8<false, false, false>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, nullptr, nullptr);
            } else if (tw_next) {
                if (in_f) hipLaunchKernelGGL((k_ntt_pass8<true, false, true>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, factors, tw_next);
                else hipLaunchKernelGGL((k_ntt_pass8<false, false, true>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, nullptr, tw_next);
            } else {
                if (in_f) hipLaunchKernelGGL((k_ntt_pass8<true, false, false>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, factors, nullptr);
                else hipLaunchKernelGGL((k_ntt_pass8<false, false, false>), grid, block, NTT8_LDS_BYTES, st, src, dst, W, a, ninv, nullptr, nullptr);
            }
        }
        src = dst;
    }
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    if (coset_gen && inverse && !big) {
        const host::el gi = host::inv(host::FR, host::el_load(coset_gen));
        PLK_TRY(scale_powers(ctx, st, data, n, gi, host::one(host::FR)));
    }
    return PLK_OK;
}

static int32_t cross_enqueue(plk_ctx* ctx, hipStream_t st, uint32_t* d, uint32_t log_g, uint64_t cols, bool inverse) {
    if (log_g == 0 || cols == 0) return PLK_OK;
    const uint32_t* W;
    PLK_TRY(ntt_twiddles(ctx, log_g, &W));
    const dim3 grid((unsigned)((cols + 255) / 256)), block(256);
    prof_begin(ctx, st, "ntt");
    switch (log_g) {
        case 1: hipLaunchKernelGGL(k_ntt_cross<1>, grid, block, 0, st, d, cols, W, inverse ? 1 : 0); break;
        case 2: hipLaunchKernelGGL(k_ntt_cross<2>, grid, block, 0, st, d, cols, W, inverse ? 1 : 0); break;
        default: hipLaunchKernelGGL(k_ntt_cross<3>, grid, block, 0, st, d, cols, W, inverse ? 1 : 0); break;
    }
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

// One half of the four-step transform of N = 2^log_n points over `world` ranks (see plk_ntt_sharded_dev in include/plonk_hip.h) on
// this rank's M = N / world elements.  Coefficients are cyclic over the ranks (rank g holds c[g + G m]), evaluations sit in column
// blocks (rank r holds X[(r M/G + k2) + M k1] at k1 M/G + k2).  With i = g + G m and k = k2 + M k1:
//   X[k2 + M k1] = sum_g w_G^(g k1) [ h^g w_N^(g k2) sum_m c_g[m] (h^G)^m w_M^(m k2) ]
// forward, before the exchange: local (coset) transform with generator h^G, then times h^g (w_N^g)^k2; after it: G-point
// transforms across the received blocks.  Inverse: the same backwards with inverse roots, 1/G on the twiddle step, 1/M on the
// local inverse transform.
int32_t ntt_sharded_half(plk_ctx* ctx, hipStream_t st, uint32_t* buf, uint32_t log_n, bool inverse, const uint64_t* coset_gen, uint32_t rank, uint32_t world,
                         int half) {
    uint32_t log_g = 0;
    while ((1u << log_g) < world) ++log_g;
    if ((1u << log_g) != world || world > 8 || rank >= world) { ctx->err = "sharded NTT: 1, 2, 4 or 8 ranks"; return PLK_ERR_INVALID_ARG; }
    if (log_n < 2 * log_g + 1 || log_n - log_g > 27) { ctx->err = "sharded NTT: transform too small or too large for this many ranks"; return PLK_ERR_INVALID_ARG; }
    const uint32_t log_m = log_n - log_g;
    const uint64_t m = 1ull << log_m;
    using namespace host;
    const el h = coset_gen ? el_load(coset_gen) : one(FR);
    if (coset_gen && is_zero(h)) { ctx->err = "coset generator is zero"; return PLK_ERR_INVERSE_OF_ZERO; }
    auto pow_u = [](el b, uint64_t e) { el r = one(FR); while (e) { if (e & 1) r = mul(FR, r, b); b = mul(FR, b, b); e >>= 1; } return r; };
    const el local = pow_u(h, world);                       // generator of the local coset transform: h^G
    const uint64_t* local_gen = coset_gen ? local.v : nullptr;
    const el w_n = root_of_unity(log_n);
    const bool trivial = !coset_gen && rank == 0;
    if (!inverse) {
        if (half == 1) {
            PLK_TRY(ntt_run(ctx, st, buf, log_m, false, local_gen));
            if (world > 1 && !trivial) PLK_TRY(scale_powers(ctx, st, buf, m, pow_u(w_n, rank), pow_u(h, rank)));
            return PLK_OK;
        }
        return cross_enqueue(ctx, st, buf, log_g, m / world, false);
    }
    if (half == 1) return cross_enqueue(ctx, st, buf, log_g, m / world, true);
    if (world > 1) {
        const el base = mul(FR, inv(FR, from_u64(FR, world)), inv(FR, pow_u(h, rank)));
        PLK_TRY(scale_powers(ctx, st, buf, m, inv(FR, pow_u(w_n, rank)), base));
    }
    return ntt_run(ctx, st, buf, log_m, true, local_gen);
}

}  // namespace plk

using namespace plk;

extern "C" int32_t plk_ntt_dev_enqueue(plk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen) {
    if (!ctx || !data_dev) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(compute_guard(ctx));
    return ntt_run(ctx, ctx->stream, (uint32_t*)data_dev, log_n, inverse != 0, coset_gen);
}
extern "C" int32_t plk_ntt_dev(plk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen) {
    PLK_TRY(plk_ntt_dev_enqueue(ctx, data_dev, log_n, inverse, coset_gen));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    prof_collect(ctx, false);
    return PLK_OK;
}
extern "C" int32_t plk_ntt(plk_ctx* ctx, uint64_t* data_host, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen) {
    if (!ctx || !data_host || log_n > 27) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)32 << log_n;
    DevBuf d;
    PLK_HIP(d.alloc_copy(data_host, bytes));
    PLK_TRY(plk_ntt_dev(ctx, d.p, log_n, inverse, coset_gen));
    PLK_HIP(hipMemcpy(data_host, d.p, bytes, hipMemcpyDeviceToHost));
    return PLK_OK;
}

extern "C" int32_t plk_scale_powers_dev(plk_ctx* ctx, void* data_dev, uint64_t n, const uint64_t base[4], const uint64_t ratio[4]) {
    if (!ctx || !base || !ratio || (n && !data_dev)) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
    if (n == 0) return PLK_OK;
    PLK_TRY(compute_guard(ctx));
    prof_begin(ctx, ctx->stream, "pointwise");
    const int32_t rc = scale_powers(ctx, ctx->stream, (uint32_t*)data_dev, n, host::el_load(ratio), host::el_load(base));
    prof_end(ctx, ctx->stream);
    return rc;
}
extern "C" int32_t plk_ntt_cross_dev(plk_ctx* ctx, void* data_dev, uint32_t log_g, uint64_t cols, int32_t inverse) {
    if (!ctx || !data_dev || log_g > 3) return PLK_ERR_INVALID_ARG;   // at most 8 row blocks (GPUs of one node)
    PLK_HIP(hipSetDevice(ctx->device));
    if (log_g == 0 || cols == 0) return PLK_OK;
    PLK_TRY(compute_guard(ctx));
    return cross_enqueue(ctx, ctx->stream, (uint32_t*)data_dev, log_g, cols, inverse != 0);
}
