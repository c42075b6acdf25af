// Integer-VALU microbenchmark for gfx950: instruction throughput of the 32-bit multiply-add family and
// of three Montgomery-product formulations (8x32-bit limbs).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct fe { uint32_t v[8]; };
__device__ constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
constexpr uint32_t INV = 0xe4866389u;

template <int OP> __global__ void __launch_bounds__(256) k_instr(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint64_t a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3, a4 = t * 11, a5 = t * 13, a6 = t * 17, a7 = t * 19;
    uint32_t m = t | 1, q = seed | 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "v"(q) : "vcc")
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 1) {  // v_mul_lo_u32
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 2) {  // v_add_u32
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_add_u32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 3) {  // v_lshl_add_u64
#define M(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(a7))
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a0);
#undef M
        } else if (OP == 4) {  // v_addc_co_u32 chain
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(l_) : "v"(m) : "vcc"); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 5) {  // v_mul_hi_u32
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 6) {  // v_mad_u32_u24
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 8) {  // v_fma_f32
#define M(x) { float f_ = __uint_as_float((uint32_t)x); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f_) : "v"(__uint_as_float(m))); x = __float_as_uint(f_); }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 9) {  // v_xor_b32
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_xor_b32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 7) {  // v_fma_f64
#define M(x) { double d_ = __longlong_as_double(x); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d_) : "v"(__longlong_as_double(a7 | 0x3ff0000000000000ull))); x = __double_as_longlong(d_); }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a0);
#undef M
        }
    }
    out[t] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 ^ a3) >> 32);
}

// (A) operand-scanning CIOS, compiler generated
__device__ __forceinline__ fe mulA(const fe& a, const fe& b) {
    uint32_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t)a.v[0] * b.v[i] + t[0];
        uint32_t A = x >> 32;
        uint32_t m = (uint32_t)x * INV;
        uint64_t y = (uint64_t)m * P[0] + (uint32_t)x;
        uint32_t Cc = y >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            x = (uint64_t)a.v[j] * b.v[i] + t[j] + A; A = x >> 32;
            y = (uint64_t)m * P[j] + (uint32_t)x + Cc; t[j - 1] = (uint32_t)y; Cc = y >> 32;
        }
        t[7] = A + Cc;
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return r;
}
// (B) product scanning, 96-bit accumulator, mad carry-out consumed by one addc (inline asm)
__device__ __forceinline__ void mac(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ fe mulB(const fe& a, const fe& b) {
    uint64_t acc = 0; uint32_t hi = 0;
    uint32_t m[8]; fe r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) mac(acc, hi, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) mac(acc, hi, m[i], P[k - i]);
        m[k] = (uint32_t)acc * INV;
        mac(acc, hi, m[k], P[0]);
        acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
#pragma unroll
        for (int i = k - 7; i < 8; ++i) mac(acc, hi, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; ++i) mac(acc, hi, m[i], P[k - i]);
        r.v[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32); hi = 0;
    }
    return r;
}
// (C) 9 limbs of 29 bits in 64-bit column accumulators: no carries at all inside a column
struct fe29 { uint32_t v[9]; };
__device__ constexpr uint32_t P29[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
constexpr uint32_t INV29 = 0x4866389u;   // -p^-1 mod 2^29 (low 29 bits of the 32-bit constant)
__device__ __forceinline__ fe29 mulC(const fe29& a, const fe29& b) {
    uint64_t acc = 0; uint32_t m[9]; fe29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * P29[k - i];
        m[k] = ((uint32_t)acc * INV29) & 0x1fffffffu;
        acc += (uint64_t)m[k] * P29[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 18; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * P29[k - i];
        r.v[k - 9] = (uint32_t)acc & 0x1fffffffu;
        acc >>= 29;
    }
    return r;
}
template <int V> __global__ void __launch_bounds__(256) k_modmul(uint32_t* io, int iters) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (V == 2) {
        fe29 a, b;
        for (int i = 0; i < 9; ++i) { a.v[i] = (io[t] * (i + 3)) & 0x0fffffffu; b.v[i] = (io[t] * (2 * i + 7)) & 0x0fffffffu; }
        for (int k = 0; k < iters; ++k) a = mulC(a, b);
        uint32_t x = 0; for (int i = 0; i < 9; ++i) x ^= a.v[i]; io[t] = x;
    } else {
        fe a, b;
        for (int i = 0; i < 8; ++i) { a.v[i] = (io[t] * (i + 3)) & 0x0fffffffu; b.v[i] = (io[t] * (2 * i + 7)) & 0x0fffffffu; }
        for (int k = 0; k < iters; ++k) a = (V == 0) ? mulA(a, b) : mulB(a, b);
        uint32_t x = 0; for (int i = 0; i < 8; ++i) x ^= a.v[i]; io[t] = x;
    }
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    const int blocks = pr.multiProcessorCount * 8, threads = 256; const size_t nthr = (size_t)blocks * threads;
    uint32_t* d; CK(hipMalloc(&d, nthr * 4)); CK(hipMemset(d, 0x5a, nthr * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[10] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_add_u32", "v_lshl_add_u64", "v_addc_co_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_fma_f64", "v_fma_f32", "v_xor_b32"};
    for (int op = 0; op < 10; ++op) {
        const int iters = 65536; float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            switch (op) {
                case 0: k_instr<0><<<blocks, threads>>>(d, iters, 1); break; case 1: k_instr<1><<<blocks, threads>>>(d, iters, 1); break;
                case 2: k_instr<2><<<blocks, threads>>>(d, iters, 1); break; case 3: k_instr<3><<<blocks, threads>>>(d, iters, 1); break;
                case 4: k_instr<4><<<blocks, threads>>>(d, iters, 1); break; case 5: k_instr<5><<<blocks, threads>>>(d, iters, 1); break;
                case 6: k_instr<6><<<blocks, threads>>>(d, iters, 1); break; case 7: k_instr<7><<<blocks, threads>>>(d, iters, 1); break;
                case 8: k_instr<8><<<blocks, threads>>>(d, iters, 1); break; default: k_instr<9><<<blocks, threads>>>(d, iters, 1); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        double ops = (double)nthr * iters * 8;
        printf("%-16s %8.3f ms  %8.2f Gop/s (lane-ops)  = %.2f lanes/clk/CU at 2.4GHz\n", names[op], ms, ops / ms * 1e-6, ops / (ms * 1e-3) / 2.4e9 / pr.multiProcessorCount);
    }
    const char* mn[3] = {"modmul CIOS (compiler)", "modmul prod-scan asm", "modmul 9x29 columns"};
    for (int occ = 0; occ < 2; ++occ) {
        int bl = occ == 0 ? blocks : pr.multiProcessorCount * 2;
        for (int v = 0; v < 3; ++v) {
            const int iters = 20000; float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (v == 0) k_modmul<0><<<bl, threads>>>(d, iters); else if (v == 1) k_modmul<1><<<bl, threads>>>(d, iters); else k_modmul<2><<<bl, threads>>>(d, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double n = (double)bl * threads * iters;
            printf("%-24s blocks/CU=%d %8.3f ms  %8.2f G modmul/s\n", mn[v], bl / pr.multiProcessorCount, ms, n / ms * 1e-6);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
