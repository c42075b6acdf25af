// Issue rate on gfx950 of the instructions that surround the multiply-adds of the 9x29-bit field product: 64-bit shifts,
// funnel shifts, masks, moves, selects.  Eight independent chains per lane; measured at 8 and at 2 waves per SIMD (the
// accumulation kernel runs at 2).  Build: make -C tools shift_microbench; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int OP> __global__ void __launch_bounds__(256) k_instr(uint32_t* out, int iters, uint32_t seed) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint64_t a0 = t + seed, a1 = t * 3 + 1, a2 = t * 5 + 2, a3 = t * 7 + 3, a4 = t * 11, a5 = t * 13, a6 = t * 17, a7 = t * 19;
    uint32_t m = t | 1, q = seed | 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "v"(q) : "vcc")
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 1) {
#define M(x) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(x))
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 2) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 3) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_and_b32 %0, %1, %0" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 4) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(l_)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 5) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_mov_b32 %0, %1" : "=v"(l_) : "v"(l_)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 6) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(l_)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 7) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(l_) : "v"(m), "v"(q)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 8) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(l_) : "v"(m) : "vcc"); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 9) {
#define M(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(a7))
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a0);
#undef M
        } else if (OP == 10) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_sub_u32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 11) {
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(l_) : "v"(m)); x = l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 12) {   // the pair a column of the product ends with today
#define M(x) { uint32_t l_ = (uint32_t)x; asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(l_)); asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(x)); x ^= l_; }
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        } else if (OP == 13) {   // mad whose multiplier is an SGPR (limb of the modulus)
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(m), "s"(seed) : "vcc")
            M(a0); M(a1); M(a2); M(a3); M(a4); M(a5); M(a6); M(a7);
#undef M
        }
    }
    out[t] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 ^ a3) >> 32) ^ q;
}

template <int OP> static int run(const char* name, uint32_t* d, int cus, int blocks_per_cu, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 32768, blocks = cus * blocks_per_cu;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_instr<OP><<<blocks, 256>>>(d, iters, 1);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double ops = (double)blocks * 256 * iters * 8;
    printf("%-28s waves/SIMD=%d %8.3f ms %9.2f G lane-ops/s  %.2f cycles per wave instruction at 2.4 GHz\n", name, blocks_per_cu, ms, ops / ms * 1e-6,
           (double)cus * 4 * 64 * 2.4e9 / (ops / (ms * 1e-3)));
    return 0;
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s CUs %d clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    const int cus = pr.multiProcessorCount;
    uint32_t* d; CK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4)); CK(hipMemset(d, 0x5a, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int bpc : {8, 2}) {
        if (run<0>("v_mad_u64_u32", d, cus, bpc, e0, e1)) return 1;
        if (run<13>("v_mad_u64_u32 (sgpr operand)", d, cus, bpc, e0, e1)) return 1;
        if (run<1>("v_lshrrev_b64", d, cus, bpc, e0, e1)) return 1;
        if (run<9>("v_lshl_add_u64", d, cus, bpc, e0, e1)) return 1;
        if (run<2>("v_alignbit_b32", d, cus, bpc, e0, e1)) return 1;
        if (run<3>("v_and_b32", d, cus, bpc, e0, e1)) return 1;
        if (run<4>("v_lshrrev_b32", d, cus, bpc, e0, e1)) return 1;
        if (run<5>("v_mov_b32", d, cus, bpc, e0, e1)) return 1;
        if (run<6>("v_bfe_u32", d, cus, bpc, e0, e1)) return 1;
        if (run<7>("v_add3_u32", d, cus, bpc, e0, e1)) return 1;
        if (run<8>("v_cndmask_b32", d, cus, bpc, e0, e1)) return 1;
        if (run<10>("v_sub_u32", d, cus, bpc, e0, e1)) return 1;
        if (run<11>("v_mul_lo_u32", d, cus, bpc, e0, e1)) return 1;
        if (run<12>("v_and_b32 + v_lshrrev_b64", d, cus, bpc, e0, e1)) return 1;
    }
    CK(hipDeviceSynchronize());
    return 0;
}
