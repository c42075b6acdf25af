// Register-resident throughput of the curve kernels' building blocks on gfx950: fp29 mul / sqr and the XYZZ mixed addition.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../plonk_amd/csrc/g1.cuh"
using namespace plk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int V> __global__ void __launch_bounds__(256) k(uint32_t* io, int iters) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    fe a = fe_const(FqP::ONE), b = fe_const(FqP::R2);
    a.v[0] ^= io[t] & 0xffff; b.v[1] ^= io[t] & 0xfff;
    if (V == 0) { for (int i = 0; i < iters; ++i) a = mul<FqP>(a, b); }
    else if (V == 1) { for (int i = 0; i < iters; ++i) a = sqr<FqP>(a); }
    else {
        aff q; q.x = a; q.y = b;
        xyzz acc = xyzz_dbl_affine(q);
        for (int i = 0; i < iters; ++i) { xyzz_madd(acc, q); q.x.v[0] ^= 1; }
        a = acc.x; a.v[1] ^= acc.zz.v[3] ^ acc.y.v[2] ^ acc.zzz.v[0];
    }
    uint32_t x = 0; for (int i = 0; i < 9; ++i) x ^= a.v[i]; io[t] = x;
}
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    uint32_t* d; const int maxblocks = pr.multiProcessorCount * 8; CK(hipMalloc(&d, (size_t)maxblocks * 256 * 4)); CK(hipMemset(d, 0x11, (size_t)maxblocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"fp29 mul", "fp29 sqr", "xyzz madd"};
    for (int bpc = 1; bpc <= 4; bpc *= 2) for (int v = 0; v < 3; ++v) {
        const int blocks = pr.multiProcessorCount * bpc, iters = v == 2 ? 2000 : 20000; float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) k<0><<<blocks, 256>>>(d, iters); else if (v == 1) k<1><<<blocks, 256>>>(d, iters); else k<2><<<blocks, 256>>>(d, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%-10s blocks/CU=%d  %8.3f ms  %8.2f G/s\n", names[v], bpc, ms, (double)blocks * 256 * iters / ms * 1e-6);
    }
    return 0;
}
