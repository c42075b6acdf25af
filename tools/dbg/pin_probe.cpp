// Probe for the host-array upload path: cost of hipHostRegister on ordinary memory, bandwidth of plain and threaded memcpy into pinned
// memory, bandwidth of hipMemcpyAsync from pageable and from pinned memory.  Build: hipcc -O2 -o pin_probe pin_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const size_t MB = 1 << 20;
    void* dev; CK(hipMalloc(&dev, 256 * MB));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (size_t mb : {32, 128}) {
        char* p = (char*)aligned_alloc(2 * MB, mb * MB); memset(p, 1, mb * MB);
        double t0 = now(), t1, t2, t3;
        hipError_t re = hipHostRegister(p, mb * MB, hipHostRegisterDefault); t1 = now();
        if (re == hipSuccess) {
            CK(hipMemcpyAsync(dev, p, mb * MB, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); t2 = now();
            CK(hipHostUnregister(p)); t3 = now();
            printf("%zu MB: register %.3f ms, copy from registered %.3f ms (%.1f GB/s), unregister %.3f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, mb / 1024.0 / (t2 - t1), (t3 - t2) * 1e3);
        } else { printf("%zu MB: hipHostRegister fails: %s\n", mb, hipGetErrorString(re)); (void)hipGetLastError(); }
        t0 = now(); CK(hipMemcpyAsync(dev, p, mb * MB, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); t1 = now();
        printf("%zu MB: copy from pageable %.3f ms (%.1f GB/s)\n", mb, (t1 - t0) * 1e3, mb / 1024.0 / (t1 - t0));
        char* pin; CK(hipHostMalloc((void**)&pin, mb * MB, hipHostMallocDefault)); memset(pin, 2, mb * MB);
        t0 = now(); CK(hipMemcpyAsync(dev, pin, mb * MB, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); t1 = now();
        printf("%zu MB: copy from pinned %.3f ms (%.1f GB/s)\n", mb, (t1 - t0) * 1e3, mb / 1024.0 / (t1 - t0));
        for (int nt : {1, 2, 4, 8, 12}) {
            t0 = now();
            std::vector<std::thread> th;
            const size_t per = mb * MB / nt;
            for (int k = 0; k < nt; ++k) th.emplace_back([=] { memcpy(pin + k * per, p + k * per, per); });
            for (auto& t : th) t.join();
            t1 = now();
            printf("%zu MB: memcpy pageable -> pinned with %2d threads %.3f ms (%.1f GB/s)\n", mb, nt, (t1 - t0) * 1e3, mb / 1024.0 / (t1 - t0));
        }
        CK(hipHostFree(pin)); free(p);
    }
    return 0;
}
