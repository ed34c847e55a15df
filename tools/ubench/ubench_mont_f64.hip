// The one multiplier that had never been run: BN254 fq Montgomery product on 5 x 52-bit limbs held in doubles, limb products split
// into exact high and low halves by the two-FMA trick (round toward zero), R = 2^260.  Not product code: it puts a measured Gmul/s beside the
// hand-scheduled 9 x 29-bit v_mad_u64_u32 product of csrc/fe_mont_gfx950.h (150.3 Gmul/s, profiles/r02_ubench_mul.txt), semantics of the
// reference's field_impl_int128.tcc:72-137,248-255 (Montgomery product with one coarse reduction, result in [0, 2p)).
//
//   hi' = fma_rz(x, y, 2^104)            in [2^104, 2^105): its 52 mantissa bits ARE floor(x y / 2^52)
//   lo' = fma_rz(x, y, 2^104 + 2^52 - hi') in [2^52, 2^53): its 52 mantissa bits ARE x y mod 2^52          (x, y < 2^52 integers held as doubles)
// The IEEE bit patterns are added into 64-bit integer column sums (the exponent fields are multiples of 2^52: they are subtracted as one
// constant per column and never disturb a column's low 52 bits); the quotient digit of the word-by-word reduction is the low half of one more
// product.  Per 256-bit product: 25 + 5 + 25 limb products of 2 FMAs + 1 exact subtraction + 2 integer additions each.
//
// Checked bit for bit against csrc/fe.hpp's mul() (R = 2^261: mont260(a, b) = 2 mont261(a, b) mod p) on 2^20 random and 4096 edge operand pairs,
// conversions from / to the 4 x u64 memory format included; timed as dependent chains like tools/ubench/ubench_mul2.hip at 1 .. 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../barretenberg_amd/csrc/fe.hpp"
using namespace bbgpu;
typedef unsigned __int128 u128;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

static const uint64_t hP64[4] = { 0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL };
struct F5 { double d[5]; };
struct Consts { double p[5]; double pinv; }; // p in 52-bit limbs, -p^-1 mod 2^52, as doubles (kernel arguments: SGPR pairs)
constexpr uint64_t M52 = (1ULL << 52) - 1;
constexpr uint64_t E_LO = 0x433ULL << 52; // exponent field of a double in [2^52, 2^53)
constexpr uint64_t E_HI = 0x467ULL << 52; // ... in [2^104, 2^105)

__device__ __forceinline__ uint64_t bits(double x) { return (uint64_t)__double_as_longlong(x); }
__device__ __forceinline__ double from_u52(uint64_t v) { return __longlong_as_double((long long)(v | E_LO)) - 0x1p52; } // exact for v < 2^52
__device__ __forceinline__ void set_round_toward_zero_f64()
{
    // MODE register, bits 3:2 = rounding of f64 / f16: 3 = toward zero (hwreg id 1, offset 2, width 2)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");
}
// number of (i, j) in [0, 5)^2 with i + j = k
__host__ __device__ constexpr int npairs(int k) { return k < 0 || k > 8 ? 0 : (k < 5 ? k + 1 : 9 - k); }

// r = a b 2^-260 mod p, r < 2p for a, b < 2p; limbs of a, b, r below 2^52.  Round-toward-zero mode must be set.
__device__ __forceinline__ void mont260(const F5& a, const F5& b, F5& r, const Consts& K)
{
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    uint64_t T[10];
#pragma unroll
    for (int k = 0; k < 10; k++) T[k] = 0 - (2 * (uint64_t)npairs(k) * E_LO + 2 * (uint64_t)npairs(k - 1) * E_HI); // every exponent field this column will receive
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = __builtin_fma(a.d[i], b.d[j], C1);
            const double lo = __builtin_fma(a.d[i], b.d[j], C2 - hi);
            T[i + j + 1] += bits(hi);
            T[i + j] += bits(lo);
        }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const double t0 = from_u52(T[i] & M52);
        const double qh = __builtin_fma(t0, K.pinv, C1);
        const double q = __builtin_fma(t0, K.pinv, C2 - qh) - 0x1p52; // (t0 * pinv) mod 2^52
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double hi = __builtin_fma(q, K.p[j], C1);
            const double lo = __builtin_fma(q, K.p[j], C2 - hi);
            T[i + j + 1] += bits(hi);
            T[i + j] += bits(lo);
        }
        T[i + 1] += T[i] >> 52; // T[i] is 0 mod 2^52 now, with all its exponent fields cancelled
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        r.d[k] = from_u52(T[5 + k] & M52);
        if (k < 4) T[6 + k] += T[5 + k] >> 52;
    }
}
// the 4 x u64 memory format (value < 2^256) <-> five 52-bit limbs in doubles
__device__ __forceinline__ void load_f5(const uint64_t* w, F5& x)
{
    x.d[0] = from_u52(w[0] & M52);
    x.d[1] = from_u52(((w[0] >> 52) | (w[1] << 12)) & M52);
    x.d[2] = from_u52(((w[1] >> 40) | (w[2] << 24)) & M52);
    x.d[3] = from_u52(((w[2] >> 28) | (w[3] << 36)) & M52);
    x.d[4] = from_u52(w[3] >> 16);
}
__device__ __forceinline__ void store_f5(const F5& x, uint64_t* w)
{
    uint64_t l[5];
#pragma unroll
    for (int i = 0; i < 5; i++) l[i] = bits(x.d[i] + 0x1p52) & M52; // exact: limb < 2^52
    w[0] = l[0] | (l[1] << 52);
    w[1] = (l[1] >> 12) | (l[2] << 40);
    w[2] = (l[2] >> 24) | (l[3] << 28);
    w[3] = (l[3] >> 36) | (l[4] << 16);
}

#define NCHAIN 512
// MODE 0: chain in limb form (what ubench_mul2 times for the 9 x 29 product); MODE 1: every product converts from and to the memory format
template <int MODE> __global__ void __launch_bounds__(256) k_chain(uint64_t* io, Consts K)
{
    set_round_toward_zero_f64();
    const size_t g = blockIdx.x * blockDim.x + threadIdx.x;
    F5 x, y;
    load_f5(io + 8 * g, x);
    load_f5(io + 8 * g + 4, y);
    for (int it = 0; it < NCHAIN; it++) {
        F5 t;
        mont260(x, y, t, K);
        if (MODE == 1) {
            uint64_t w[4];
            store_f5(t, w);
            asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])); // keep both conversions in the loop
            load_f5(w, t);
        }
        y = x;
        x = t;
    }
    store_f5(x, io + 8 * g);
}
// one product per thread by both multipliers: out[0..3] = f64 path, out[4..7] = fe.hpp's 9 x 29 path (R = 2^261), both packed as 4 x u64 (values < 2p)
__global__ void __launch_bounds__(256) k_check(const uint64_t* in, uint64_t* out, size_t n, Consts K)
{
    const size_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    uint32_t wa[8], wb[8];
    for (int i = 0; i < 4; i++) {
        wa[2 * i] = (uint32_t)in[8 * g + i]; wa[2 * i + 1] = (uint32_t)(in[8 * g + i] >> 32);
        wb[2 * i] = (uint32_t)in[8 * g + 4 + i]; wb[2 * i + 1] = (uint32_t)(in[8 * g + 4 + i] >> 32);
    }
    const auto ref = mul(unpack<FqP>(wa), unpack<FqP>(wb)); // before the rounding mode changes (integer code: indifferent to it anyway)
    uint32_t wr[8];
    pack(ref, wr);
    for (int i = 0; i < 4; i++) out[8 * g + 4 + i] = (uint64_t)wr[2 * i] | ((uint64_t)wr[2 * i + 1] << 32);
    set_round_toward_zero_f64();
    F5 a, b, r;
    load_f5(in + 8 * g, a);
    load_f5(in + 8 * g + 4, b);
    mont260(a, b, r, K);
    store_f5(r, out + 8 * g);
}

// ---- host: 256-bit helpers -------------------------------------------------------------------------------------------------------------
static int ge4(const uint64_t a[4], const uint64_t b[4]) { for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return 1; }
static void sub4(uint64_t a[4], const uint64_t b[4]) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } }
static void canon(uint64_t a[4]) { while (ge4(a, hP64)) sub4(a, hP64); }
static void dbl_mod(uint64_t a[4]) { canon(a); uint64_t c = 0; for (int i = 0; i < 4; i++) { uint64_t n = a[i] >> 63; a[i] = (a[i] << 1) | c; c = n; } canon(a); } // p < 2^254: no overflow
static uint64_t sm = 0x9e3779b97f4a7c15ULL;
static uint64_t splitmix() { uint64_t z = (sm += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
    Consts K;
    { // p in 52-bit limbs; -p^-1 mod 2^52 by Newton iteration on the low limb
        const uint64_t l[5] = { hP64[0] & M52, ((hP64[0] >> 52) | (hP64[1] << 12)) & M52, ((hP64[1] >> 40) | (hP64[2] << 24)) & M52, ((hP64[2] >> 28) | (hP64[3] << 36)) & M52, hP64[3] >> 16 };
        for (int i = 0; i < 5; i++) K.p[i] = (double)l[i];
        uint64_t inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - l[0] * inv;
        K.pinv = (double)((0 - inv) & M52);
    }
    // ---- bit check: 2^20 random pairs below 2p + 4096 pairs from a grid of edge values
    uint64_t p2[4]; { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)hP64[i] * 2; p2[i] = (uint64_t)c; c >>= 64; } }
    std::vector<std::vector<uint64_t>> edge;
    auto push = [&](uint64_t a, uint64_t b, uint64_t c, uint64_t d) { uint64_t v[4] = { a, b, c, d }; if (!ge4(v, p2)) edge.push_back({ a, b, c, d }); };
    push(0, 0, 0, 0); push(1, 0, 0, 0); push(2, 0, 0, 0); push(M52, 0, 0, 0); push(M52 + 1, 0, 0, 0); push(~0ULL, 0, 0, 0); push(0, 1, 0, 0);
    push(~0ULL, ~0ULL, 0, 0); push(~0ULL, ~0ULL, ~0ULL, 0); push(~0ULL, ~0ULL, ~0ULL, 0x0fffffffffffffffULL); push(0, 0, 0, 0x3000000000000000ULL);
    push(~0ULL, ~0ULL, ~0ULL, 0x3fffffffffffffffULL); push(0, 0, 0, 1ULL << 16); push(~0ULL << 52, 0xfff, 0, 0); push(0xfffffffffffff000ULL, 0xffffffffff, 0, 0);
    for (int d = -2; d <= 2; d++) { // p + d, 2p + d (below 2p), 2^k boundaries of the 52-bit limbs
        uint64_t v[4]; memcpy(v, hP64, 32); if (d < 0) { uint64_t m[4] = { (uint64_t)-d, 0, 0, 0 }; sub4(v, m); } else v[0] += (uint64_t)d; push(v[0], v[1], v[2], v[3]);
        memcpy(v, p2, 32); uint64_t m[4] = { (uint64_t)(3 - d), 0, 0, 0 }; sub4(v, m); push(v[0], v[1], v[2], v[3]);
    }
    for (int k = 1; k < 5; k++) { uint64_t v[4] = { 0, 0, 0, 0 }; const int bit = 52 * k; v[bit >> 6] = 1ULL << (bit & 63); push(v[0], v[1], v[2], v[3]); uint64_t one[4] = { 1, 0, 0, 0 }; sub4(v, one); push(v[0], v[1], v[2], v[3]); }
    while (edge.size() < 64) { uint64_t v[4] = { splitmix(), splitmix(), splitmix(), splitmix() & 0x3fffffffffffffffULL }; if (!ge4(v, p2)) edge.push_back({ v[0], v[1], v[2], v[3] }); }
    edge.resize(64);
    const size_t nr = (size_t)1 << 20, n = nr + 64 * 64;
    std::vector<uint64_t> in(n * 8), out(n * 8);
    for (size_t g = 0; g < nr; g++)
        for (int h = 0; h < 2; h++) {
            uint64_t v[4];
            do { for (int i = 0; i < 4; i++) v[i] = splitmix(); v[3] &= 0x7fffffffffffffffULL; } while (ge4(v, p2)); // uniform below 2p
            memcpy(&in[8 * g + 4 * h], v, 32);
        }
    for (size_t e = 0; e < 64 * 64; e++) { memcpy(&in[8 * (nr + e)], edge[e / 64].data(), 32); memcpy(&in[8 * (nr + e) + 4], edge[e % 64].data(), 32); }
    uint64_t *d_in, *d_out;
    CHECK(hipMalloc(&d_in, in.size() * 8)); CHECK(hipMalloc(&d_out, out.size() * 8));
    CHECK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    k_check<<<(unsigned)((n + 255) / 256), 256>>>(d_in, d_out, n, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, above_2p = 0;
    for (size_t g = 0; g < n; g++) {
        uint64_t a[4], b[4];
        memcpy(a, &out[8 * g], 32); memcpy(b, &out[8 * g + 4], 32);
        if (ge4(a, p2)) above_2p++;
        canon(a); dbl_mod(b); // mont260 = 2 * mont261 (mod p)
        if (memcmp(a, b, 32)) { if (bad < 4) printf("mismatch at pair %zu\n", g); bad++; }
    }
    printf("bit check against fe.hpp mul(): %zu operand pairs (2^20 random below 2p + 4096 edge pairs), mismatches %zu, results >= 2p: %zu\n", n, bad, above_2p);
    hipFree(d_in); hipFree(d_out);
    // ---- throughput, dependent chains
    for (int mode = 0; mode < 2; mode++)
        for (int wps : { 1, 2, 3, 4 }) {
            const size_t nt = (size_t)cus * wps * 256;
            std::vector<uint64_t> h(nt * 8);
            for (size_t i = 0; i < nt * 2; i++) { uint64_t v[4]; do { for (int j = 0; j < 4; j++) v[j] = splitmix(); v[3] &= 0x3fffffffffffffffULL; } while (ge4(v, p2)); memcpy(&h[4 * i], v, 32); }
            uint64_t* d; CHECK(hipMalloc(&d, h.size() * 8));
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms;
            for (int rep = 0; rep < 2; rep++) { // the second run is the timed one
                CHECK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
                CHECK(hipEventRecord(e0));
                if (mode == 0) k_chain<0><<<cus * wps, 256>>>(d, K); else k_chain<1><<<cus * wps, 256>>>(d, K);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("w/SIMD %d  %-34s %8.3f ms  %8.2f Gmul/s\n", wps, mode == 0 ? "f64 5x52 product, limb form" : "f64 5x52 product + both conversions", ms, (double)nt * NCHAIN / ms / 1e6);
            hipFree(d);
        }
    return 0;
}
