// Known-traffic kernels that calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the hot kernels
// (MI355X_MICROARCH.md, section HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   calib_gather64 : every lane reads ONE random 64-byte row (4 x global_load_dwordx4) of a 2 GiB table -- msm_accumulate_kernel's gather
//   calib_stream32 : every lane reads 32 contiguous bytes (2 x dwordx4) and writes 32                  -- ntt_pass_kernel's load / store
//   calib_stream16 : every lane reads 16 contiguous bytes and writes 16                                -- the guide's x2 case
// Run once plain (prints the known byte counts), then under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes);
// tools/pmc_summary.py divides.  Not product code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void __launch_bounds__(256) calib_gather64(const uint4* __restrict__ table, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out, uint32_t rows)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows) return;
    const uint4* r = table + (size_t)idx[t] * 4;
    const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
    const uint32_t v = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    if (v == 0x12345678u) out[0] = v; // data-dependent, practically never taken: keeps the loads, writes nothing
}
__global__ void __launch_bounds__(256) calib_stream32(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint4 a = in[2 * (size_t)t], b = in[2 * (size_t)t + 1];
    out[2 * (size_t)t] = make_uint4(a.x + 1, a.y, a.z, a.w);
    out[2 * (size_t)t + 1] = make_uint4(b.x + 1, b.y, b.z, b.w);
}
__global__ void __launch_bounds__(256) calib_stream16(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint4 a = in[t];
    out[t] = make_uint4(a.x + 1, a.y, a.z, a.w);
}

int main()
{
    const size_t table_bytes = (size_t)2 << 30; // far beyond the 256 MiB Infinity Cache
    const uint32_t table_rows = (uint32_t)(table_bytes / 64), rows = 1u << 24; // 2^24 gathered rows = 1 GiB of payload
    uint4* table; uint32_t *idx, *out;
    CHECK(hipMalloc(&table, table_bytes)); CHECK(hipMemset(table, 1, table_bytes));
    std::vector<uint32_t> h(rows);
    uint64_t st = 0x9e3779b97f4a7c15ULL;
    for (auto& v : h) { st += 0x9e3779b97f4a7c15ULL; uint64_t z = st; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; z ^= z >> 31; v = (uint32_t)(z % table_rows); }
    CHECK(hipMalloc(&idx, (size_t)rows * 4)); CHECK(hipMemcpy(idx, h.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&out, 64));
    const uint32_t n = 1u << 24; // stream kernels: 2^24 lanes
    uint4 *sin, *sout;
    CHECK(hipMalloc(&sin, (size_t)n * 32)); CHECK(hipMalloc(&sout, (size_t)n * 32)); CHECK(hipMemset(sin, 2, (size_t)n * 32));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0)); calib_gather64<<<rows / 256, 256>>>(table, idx, out, rows); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("calib_gather64 known_read_bytes %zu (rows * 64) + index %zu  known_write_bytes 0  %.3f ms  %.1f GB/s payload\n", (size_t)rows * 64, (size_t)rows * 4, ms, rows * 64.0 / ms / 1e6);
        CHECK(hipEventRecord(e0)); calib_stream32<<<n / 256, 256>>>(sin, sout, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("calib_stream32 known_read_bytes %zu known_write_bytes %zu  %.3f ms  %.1f GB/s read+write\n", (size_t)n * 32, (size_t)n * 32, ms, n * 64.0 / ms / 1e6);
        CHECK(hipEventRecord(e0)); calib_stream16<<<n / 256, 256>>>(sin, sout, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("calib_stream16 known_read_bytes %zu known_write_bytes %zu  %.3f ms  %.1f GB/s read+write\n", (size_t)n * 16, (size_t)n * 16, ms, n * 32.0 / ms / 1e6);
    }
    return 0;
}
