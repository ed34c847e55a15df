// multiplier codegen experiments on the product header (fe.hpp): throughput of mul / sqr chains, bit-checked on host
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../barretenberg_amd/csrc/fe.hpp"
using namespace bbgpu;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define NCHAIN 512
template <int MODE> __global__ void __launch_bounds__(256) k(uint32_t* io)
{
    const size_t g = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<FqP, 1, 2> x, y;
    for (int i = 0; i < 9; i++) { x.d[i] = io[(2 * g) * 9 + i]; y.d[i] = io[(2 * g + 1) * 9 + i]; }
    for (int it = 0; it < NCHAIN; it++) {
        if (MODE == 0) { auto t = mul(x, y); y = x; x = t; }
        else { auto t = sqr(x); x = mul(t, y); it++; }
    }
    for (int i = 0; i < 9; i++) io[(2 * g) * 9 + i] = x.d[i];
}
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); int cus = prop.multiProcessorCount;
    for (int wps : {1, 2, 3, 4}) {
        size_t nt = (size_t)cus * wps * 256;
        std::vector<uint32_t> h(nt * 18);
        uint64_t st = 12345; for (auto& v : h) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; v = (uint32_t)(st >> 35) & 0x1fffffff; }
        for (size_t i = 0; i < nt * 2; i++) h[i * 9 + 8] &= 0xfffff;
        uint32_t* d; CHECK(hipMalloc(&d, h.size() * 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms;
        for (int mode = 0; mode < 2; mode++) {
            CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            if (mode == 0) k<0><<<cus * wps, 256>>>(d); else k<1><<<cus * wps, 256>>>(d);
            CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipEventRecord(e0)); if (mode == 0) k<0><<<cus * wps, 256>>>(d); else k<1><<<cus * wps, 256>>>(d);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint32_t> o(18 * 64); CHECK(hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int t = 0; t < 64; t++) { Fe<FqP, 1, 2> x, y; for (int i = 0; i < 9; i++) { x.d[i] = h[(2 * t) * 9 + i]; y.d[i] = h[(2 * t + 1) * 9 + i]; }
                for (int it = 0; it < NCHAIN; it++) { if (mode == 0) { auto tt = mul(x, y); y = x; x = tt; } else { auto tt = sqr(x); x = mul(tt, y); it++; } }
                for (int i = 0; i < 9; i++) bad += x.d[i] != o[(2 * t) * 9 + i]; }
            printf("w/SIMD %d  %-8s %8.3f ms  %8.2f Gmul/s  mismatches %d\n", wps, mode == 0 ? "mul" : "sqr+mul", ms, (double)nt * NCHAIN / ms / 1e6, bad);
        }
        hipFree(d);
    }
    return 0;
}
