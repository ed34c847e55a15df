// ubench_pcie.hip -- what the host-pointer entries (bbgpu_ntt, bbgpu_msm_g1) can expect from the link on this box:
// 32 MiB host <-> device with pageable memory, with the same memory registered (hipHostRegister) and with hipHostMalloc memory;
// the cost of registering / unregistering; both directions at once; a kernel reading / writing registered host memory directly.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_pcie ubench_pcie.hip && ./ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static double med(F f, int reps = 9) { std::vector<double> t; f(); for (int i = 0; i < reps; i++) { double a = now(); f(); t.push_back(now() - a); } std::sort(t.begin(), t.end()); return t[t.size() / 2]; }
__global__ void k_read(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; const size_t s = (size_t)gridDim.x * blockDim.x; for (; i + s < n16; i += 2 * s) { uint4 a = src[i], b = src[i + s]; dst[i] = a; dst[i + s] = b; } }
int main()
{
    const size_t B = 32u << 20;
    void* d = nullptr; void* d2 = nullptr;
    CHK(hipMalloc(&d, B)); CHK(hipMalloc(&d2, B));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char* pg = (char*)aligned_alloc(4096, B); memset(pg, 1, B);
    char* pg2 = (char*)aligned_alloc(4096, B); memset(pg2, 2, B);
    void* pin = nullptr; CHK(hipHostMalloc(&pin, B, hipHostMallocDefault)); memset(pin, 3, B);
    printf("32 MiB transfers, median of 9 (ms | GB/s)\n");
    auto rep = [&](const char* what, double ms, double bytes) { printf("  %-58s %7.3f ms  %6.1f GB/s\n", what, ms, bytes / ms / 1e6); };
    rep("pageable H2D (hipMemcpyAsync + sync)", med([&] { hipMemcpyAsync(d, pg, B, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }), B);
    rep("pageable D2H", med([&] { hipMemcpyAsync(pg, d, B, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); }), B);
    rep("hipHostMalloc H2D", med([&] { hipMemcpyAsync(d, pin, B, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }), B);
    rep("hipHostMalloc D2H", med([&] { hipMemcpyAsync(pin, d, B, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); }), B);
    rep("hipHostRegister + hipHostUnregister (32 MiB, touched)", med([&] { hipHostRegister(pg, B, hipHostRegisterDefault); hipHostUnregister(pg); }), B);
    { double a = now(); CHK(hipHostRegister(pg, B, hipHostRegisterDefault)); printf("  hipHostRegister alone %.3f ms\n", now() - a); }
    CHK(hipHostRegister(pg2, B, hipHostRegisterDefault));
    rep("registered H2D", med([&] { hipMemcpyAsync(d, pg, B, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }), B);
    rep("registered D2H", med([&] { hipMemcpyAsync(pg, d, B, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); }), B);
    rep("registered H2D + D2H at once (two streams, 64 MiB)", med([&] { hipMemcpyAsync(d, pg, B, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pg2, d2, B, hipMemcpyDeviceToHost, s2); hipStreamSynchronize(s1); hipStreamSynchronize(s2); }), 2.0 * B);
    rep("registered H2D in 8 chunks of 4 MiB", med([&] { for (int c = 0; c < 8; c++) hipMemcpyAsync((char*)d + c * (B / 8), pg + c * (B / 8), B / 8, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }), B);
    void* dp = nullptr; CHK(hipHostGetDevicePointer(&dp, pg, 0)); void* dp2 = nullptr; CHK(hipHostGetDevicePointer(&dp2, pg2, 0));
    for (int blocks : { 256, 1024, 4096 }) {
        char nm[96];
        snprintf(nm, sizeof nm, "kernel reads registered host memory -> HBM (%d x 256 lanes)", blocks);
        rep(nm, med([&] { k_read<<<blocks, 256, 0, s1>>>((const uint4*)dp, (uint4*)d, B / 16); hipStreamSynchronize(s1); }), B);
        snprintf(nm, sizeof nm, "kernel writes HBM -> registered host memory (%d x 256 lanes)", blocks);
        rep(nm, med([&] { k_read<<<blocks, 256, 0, s1>>>((const uint4*)d, (uint4*)dp2, B / 16); hipStreamSynchronize(s1); }), B);
    }
    { double a = now(); CHK(hipHostUnregister(pg)); printf("  hipHostUnregister alone %.3f ms\n", now() - a); }
    CHK(hipHostUnregister(pg2));
    // FIRST use of a fresh host range (what a caller's new polynomial looks like to the runtime): mmap'd, written by the CPU, then copied
    printf("first and second copy from a fresh, CPU-written host range (ms)\n");
    for (size_t sz : { (size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20 }) {
        for (int mode = 0; mode < 3; mode++) {
            char* f = (char*)aligned_alloc(4096, sz);
            memset(f, 5, sz);
            double r0 = 0;
            if (mode == 1) { double a = now(); CHK(hipHostRegister(f, sz, hipHostRegisterDefault)); r0 = now() - a; }
            double a = now(); hipMemcpyAsync(d, f, sz, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); double t1 = now() - a;
            a = now(); hipMemcpyAsync(d, f, sz, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); double t2 = now() - a;
            a = now(); hipMemcpyAsync(f, d, sz, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); double t3 = now() - a;
            if (mode == 2) { // through our own pinned staging buffer, CPU memcpy first
                a = now(); memcpy(pin, f, sz); hipMemcpyAsync(d, pin, sz, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); t3 = now() - a;
            }
            printf("  %2zu MiB %-28s register %.3f  first H2D %.3f  second H2D %.3f  %s %.3f\n", sz >> 20, mode == 0 ? "pageable" : mode == 1 ? "hipHostRegister first" : "pageable, then via staging",
                   r0, t1, t2, mode == 2 ? "memcpy+H2D from pinned" : "D2H", t3);
            if (mode == 1) hipHostUnregister(f);
            free(f);
        }
    }
    // a caller that allocates its buffers anew for every job (a prover's polynomials): allocate, write, copy, free -- ten rounds
    printf("allocate + write + H2D + free, ten rounds (ms per copy call; the address the allocator hands out)\n");
    for (int mode = 0; mode < 3; mode++) {
        printf("  %-52s", mode == 0 ? "8 MiB pageable, hipMemcpyAsync:" : mode == 1 ? "8 MiB, hipHostRegister / copy / hipHostUnregister:" : "8 MiB, CPU memcpy into own pinned buffer, copy:");
        for (int r = 0; r < 10; r++) {
            const size_t sz = (size_t)8 << 20;
            char* f = (char*)aligned_alloc(64, sz);
            memset(f, r, sz);
            double a = now();
            if (mode == 1) hipHostRegister(f, sz, hipHostRegisterDefault);
            if (mode == 2) memcpy(pin, f, sz);
            hipMemcpyAsync(d, mode == 2 ? pin : f, sz, hipMemcpyHostToDevice, s1);
            double t = now() - a;
            hipStreamSynchronize(s1);
            if (mode == 1) hipHostUnregister(f);
            printf(" %.3f", t);
            if (r == 9) printf("   (%p)", (void*)f);
            free(f);
        }
        printf("\n");
    }
    // a small transfer for the fixed cost
    rep("pageable H2D 4 KiB", med([&] { hipMemcpyAsync(d, pg, 4096, hipMemcpyHostToDevice, s1); hipStreamSynchronize(s1); }), 4096);
    return 0;
}
