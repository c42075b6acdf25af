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
// data[i] *= base * g^i.  pow2[k] = g^(2^k), base: x*2^256 words.  Each thread walks 8 consecutive elements.
__global__ void __launch_bounds__(256) k_scale_powers(uint32_t* __restrict__ data, uint64_t n, const uint32_t* __restrict__ pow2, const uint32_t* __restrict__ base) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 >= n) return;
    const fe to_rp = fe_const(FrP::TO_RP);
    fe cur = mul<FrP>(ld_fe(base), to_rp);
    for (int k = 0; k < 40; ++k)
        if ((i0 >> k) & 1) cur = mul<FrP>(cur, mul<FrP>(ld_fe(pow2 + 8 * k), to_rp));
    const fe g = mul<FrP>(ld_fe(pow2), to_rp);
    for (int k = 0; k < 8 && i0 + k < n; ++k) {
        st_fe<FrP>(data + 8 * (i0 + k), mul<FrP>(ld_fe(data + 8 * (i0 + k)), cur));
        cur = mul<FrP>(cur, g);
    }
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
    hipFree(ctx->ntt_tmp[0]); hipFree(ctx->ntt_tmp[1]);
    ctx->ntt_tmp[0] = ctx->ntt_tmp[1] = nullptr; ctx->ntt_tmp_cap = 0;
}

static int32_t scale_powers(plk_ctx* ctx, hipStream_t st, uint32_t* data, uint64_t n, const host::el& g, const host::el& base) {
    uint64_t buf[41 * 4];
    host::el p = g;
    for (int k = 0; k < 40; ++k) { memcpy(buf + 4 * k, p.v, 32); p = host::mul(host::FR, p, p); }
    memcpy(buf + 160, base.v, 32);
    uint32_t* dev = nullptr;
    PLK_HIP(hipMalloc(&dev, sizeof(buf)));
    PLK_HIP(hipMemcpyAsync(dev, buf, sizeof(buf), hipMemcpyHostToDevice, st));
    PLK_HIP(hipStreamSynchronize(st));   // buf is on the stack
    hipLaunchKernelGGL(k_scale_powers, dim3((unsigned)((n / 8 + 256) / 256)), dim3(256), 0, st, data, n, dev, dev + 8 * 40);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(st));
    PLK_HIP(hipFree(dev));
    return PLK_OK;
}

int32_t ntt_run(plk_ctx* ctx, hipStream_t st, uint32_t* data, uint32_t log_n, bool inverse, const uint64_t* coset_gen) {
    if (log_n > 27) { ctx->err = "NTT size above 2^27"; return PLK_ERR_INVALID_ARG; }
    const uint64_t n = 1ull << log_n;
    if (log_n == 0) return PLK_OK;
    const uint32_t* W;
    PLK_TRY(ntt_twiddles(ctx, log_n, &W));
    // pass plan
    uint32_t lr[3], lns[3], lt[3], npass;
    if (log_n <= 10) { npass = 1; lr[0] = log_n; lns[0] = 0; lt[0] = 0; }
    else {
        npass = (log_n + 9) / 10;
        const uint32_t base = log_n / npass, rem = log_n % npass;
        uint32_t acc = 0;
        for (uint32_t p = 0; p < npass; ++p) {
            lr[p] = base + (p < rem ? 1 : 0);
            lns[p] = acc;
            uint32_t t = 12 - lr[p];
            if (t > log_n - lr[p]) t = log_n - lr[p];
            if (p > 0 && t > acc) t = acc;
            lt[p] = t;
            acc += lr[p];
        }
    }
    if (npass > 1 && ctx->ntt_tmp_cap < n) {
        PLK_HIP(hipStreamSynchronize(st));
        hipFree(ctx->ntt_tmp[0]); hipFree(ctx->ntt_tmp[1]);
        ctx->ntt_tmp[0] = ctx->ntt_tmp[1] = nullptr; ctx->ntt_tmp_cap = 0;
        PLK_HIP(hipMalloc(&ctx->ntt_tmp[0], n * 32));
        PLK_HIP(hipMalloc(&ctx->ntt_tmp[1], n * 32));
        ctx->ntt_tmp_cap = n;
    }
    if (coset_gen && !inverse) PLK_TRY(scale_powers(ctx, st, data, n, host::el_load(coset_gen), host::one(host::FR)));
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
        uint32_t* dst = last ? data : ctx->ntt_tmp[p & 1];
        const uint32_t groups = (uint32_t)(n >> (lr[p] + lt[p]));
        hipLaunchKernelGGL(k_ntt_pass, dim3(groups), dim3(NTT_THREADS), 0, st, src, dst, W, log_n, lr[p], lns[p], lt[p], inverse ? 1 : 0, last ? 1 : 0,
                           (last && inverse) ? 1 : 0, ninv);
        src = dst;
    }
    prof_end(ctx, st);
    PLK_HIP(hipGetLastError());
    if (coset_gen && inverse) {
        const host::el gi = host::inv(host::FR, host::el_load(coset_gen));
        PLK_TRY(scale_powers(ctx, st, data, n, gi, host::one(host::FR)));
    }
    return PLK_OK;
}

}  // namespace plk

using namespace plk;

extern "C" int32_t plk_ntt_dev_enqueue(plk_ctx* ctx, void* data_dev, uint32_t log_n, int32_t inverse, const uint64_t* coset_gen) {
    if (!ctx || !data_dev) return PLK_ERR_INVALID_ARG;
    PLK_HIP(hipSetDevice(ctx->device));
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
    void* d = nullptr;
    PLK_HIP(hipMalloc(&d, bytes));
    PLK_HIP(hipMemcpy(d, data_host, bytes, hipMemcpyHostToDevice));
    int32_t rc = plk_ntt_dev(ctx, d, log_n, inverse, coset_gen);
    if (rc == PLK_OK && hipMemcpy(data_host, d, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = PLK_ERR_HIP;
    hipFree(d);
    return rc;
}
