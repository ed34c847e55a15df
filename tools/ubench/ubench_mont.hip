// Montgomery-multiplier variant shoot-out on gfx950 (fq of BN254). Not product code:
// decides the limb schedule for barretenberg_amd/csrc/field.cuh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned __int128 u128;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// ---- BN254 fq ----
__device__ __constant__ const uint64_t P64[4] = {0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t hP64[4] = {0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#define PINV64 0x87d20782e4866389ULL

// ---------- V1: 4x64 via __int128, coarse reduction (result < 2p) ----------
struct V1 { uint64_t d[4]; };
__host__ __device__ inline void v1_mul(const V1& a, const V1& b, V1& r, const uint64_t* P) {
  uint64_t t[5] = {0,0,0,0,0};
  #pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    #pragma unroll
    for (int j = 0; j < 4; j++) { c += (u128)a.d[i] * b.d[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    uint64_t t4 = t[4] + (uint64_t)c;
    uint64_t m = t[0] * PINV64;
    c = (u128)m * P[0] + t[0]; c >>= 64;
    #pragma unroll
    for (int j = 1; j < 4; j++) { c += (u128)m * P[j] + t[j]; t[j-1] = (uint64_t)c; c >>= 64; }
    c += t4; t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
  }
  r.d[0]=t[0]; r.d[1]=t[1]; r.d[2]=t[2]; r.d[3]=t[3];
}

// ---------- V2: 8x32 CIOS with 64-bit mads ----------
struct V2 { uint32_t d[8]; };
__device__ __constant__ const uint32_t P32[8] = {0xD87CFD47u,0x3C208C16u,0x6871ca8du,0x97816a91u,0x8181585du,0xb85045b6u,0xe131a029u,0x30644e72u};
__device__ inline void v2_mul(const V2& a, const V2& b, V2& r) {
  uint32_t t[10];
  #pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
  const uint32_t pinv = 0xe4866389u;
  #pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    #pragma unroll
    for (int j = 0; j < 8; j++) { c = (uint64_t)a.d[i] * b.d[j] + t[j] + c; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
    uint32_t m = t[0] * pinv;
    c = (uint64_t)m * P32[0] + t[0]; c >>= 32;
    #pragma unroll
    for (int j = 1; j < 8; j++) { c = (uint64_t)m * P32[j] + t[j] + c; t[j-1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
  }
  #pragma unroll
  for (int i = 0; i < 8; i++) r.d[i] = t[i];
}

// ---------- V3: 9x29 lazy-carry product scanning, R = 2^261 ----------
struct V3 { uint32_t d[9]; };
#define M29 0x1fffffffu
// p in radix 2^29 and -p^-1 mod 2^29 are filled by the host.
__device__ __constant__ uint32_t P29[9];
__device__ __constant__ uint32_t PINV29;
__device__ inline void v3_mul(const V3& a, const V3& b, V3& r) {
  uint64_t col[18];
  // schoolbook columns
  #pragma unroll
  for (int k = 0; k < 17; k++) {
    uint64_t acc = 0;
    #pragma unroll
    for (int i = 0; i < 9; i++) { int j = k - i; if (j >= 0 && j < 9) acc += (uint64_t)a.d[i] * b.d[j]; }
    col[k] = acc;
  }
  col[17] = 0;
  uint32_t m[9];
  uint64_t carry = 0;
  #pragma unroll
  for (int k = 0; k < 9; k++) {
    uint64_t acc = col[k] + carry;
    #pragma unroll
    for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P29[k - i];
    m[k] = ((uint32_t)acc * PINV29) & M29;
    acc += (uint64_t)m[k] * P29[0];
    carry = acc >> 29;
  }
  #pragma unroll
  for (int k = 9; k < 18; k++) {
    uint64_t acc = col[k] + carry;
    #pragma unroll
    for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P29[k - i];
    r.d[k - 9] = (uint32_t)acc & M29;
    carry = acc >> 29;
  }
}

// host reference for V1 (same code), conversions
static void host_chain(V1 x, V1 y, int n, V1& out) { for (int i = 0; i < n; i++) { V1 t; v1_mul(x, y, t, hP64); y = x; x = t; } out = x; }

#define NCHAIN 512
__global__ void __launch_bounds__(256) k_v1(V1* io) {
  size_t g = blockIdx.x * blockDim.x + threadIdx.x; V1 x = io[2*g], y = io[2*g+1];
  for (int i = 0; i < NCHAIN; i++) { V1 t; v1_mul(x, y, t, P64); y = x; x = t; }
  io[2*g] = x;
}
__global__ void __launch_bounds__(256) k_v2(V2* io) {
  size_t g = blockIdx.x * blockDim.x + threadIdx.x; V2 x = io[2*g], y = io[2*g+1];
  for (int i = 0; i < NCHAIN; i++) { V2 t; v2_mul(x, y, t); y = x; x = t; }
  io[2*g] = x;
}
__global__ void __launch_bounds__(256) k_v3(V3* io) {
  size_t g = blockIdx.x * blockDim.x + threadIdx.x; V3 x = io[2*g], y = io[2*g+1];
  for (int i = 0; i < NCHAIN; i++) { V3 t; v3_mul(x, y, t); y = x; x = t; }
  io[2*g] = x;
}

// ---- host bigint helpers (256-bit as 4x64) for V3 conversions ----
static void to29(const uint64_t v[5], uint32_t o[9]) { // v: up to 261 bits in 5 limbs
  for (int i = 0; i < 9; i++) { int bit = 29 * i; int w = bit >> 6, s = bit & 63; uint64_t x = v[w] >> s; if (s > 35 && w + 1 < 5) x |= v[w+1] << (64 - s); o[i] = (uint32_t)x & M29; }
}
static void from29(const uint32_t d[9], uint64_t v[5]) {
  for (int i = 0; i < 5; i++) v[i] = 0;
  for (int i = 0; i < 9; i++) { int bit = 29 * i; int w = bit >> 6, s = bit & 63; u128 x = (u128)d[i] << s; v[w] += (uint64_t)x; /* no overflow: disjoint bits when normalised */ if (w + 1 < 5) v[w+1] += (uint64_t)(x >> 64); }
}
// big-number mod p helpers via simple shift-add (slow, host only)
static int ge(const uint64_t a[5], const uint64_t b[5]) { for (int i = 4; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return 1; }
static void sub5(uint64_t a[5], const uint64_t b[5]) { u128 br = 0; for (int i = 0; i < 5; i++) { u128 t = (u128)a[i] - b[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } }
static void dblmod(uint64_t a[5], const uint64_t p[5]) { uint64_t c = 0; for (int i = 0; i < 5; i++) { uint64_t n = a[i] >> 63; a[i] = (a[i] << 1) | c; c = n; } if (ge(a, p)) sub5(a, p); }
// r = a * 2^k mod p
static void shlmod(const uint64_t a4[4], int k, uint64_t out4[4]) { uint64_t a[5] = {a4[0],a4[1],a4[2],a4[3],0}, p[5] = {hP64[0],hP64[1],hP64[2],hP64[3],0}; while (ge(a,p)) sub5(a,p); for (int i = 0; i < k; i++) dblmod(a, p); for (int i = 0; i < 4; i++) out4[i] = a[i]; }

static uint64_t sm_state = 0x9e3779b97f4a7c15ULL;
static uint64_t splitmix() { uint64_t z = (sm_state += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); int cus = prop.multiProcessorCount;
  // constants for V3
  { uint64_t p5[5] = {hP64[0],hP64[1],hP64[2],hP64[3],0}; uint32_t p29[9]; to29(p5, p29);
    uint32_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - p29[0] * inv; uint32_t pinv = (0u - inv) & M29;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(P29), p29, sizeof(p29))); CHECK(hipMemcpyToSymbol(HIP_SYMBOL(PINV29), &pinv, 4)); }
  const int wps = 4; size_t nt = (size_t)cus * wps * 256;
  std::vector<V1> h1(2 * nt);
  for (auto& v : h1) { for (int i = 0; i < 4; i++) v.d[i] = splitmix(); v.d[3] &= 0x1fffffffffffffffULL; }
  // expected for a few threads
  const int NCHK = 64; std::vector<V1> exp(NCHK); for (int i = 0; i < NCHK; i++) host_chain(h1[2*i], h1[2*i+1], NCHAIN, exp[i]);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); float ms;
  double nmul = (double)nt * NCHAIN;
  // V1
  { V1* d; CHECK(hipMalloc(&d, 2 * nt * sizeof(V1))); CHECK(hipMemcpy(d, h1.data(), 2 * nt * sizeof(V1), hipMemcpyHostToDevice));
    k_v1<<<cus * wps, 256>>>(d); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(d, h1.data(), 2 * nt * sizeof(V1), hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0)); k_v1<<<cus * wps, 256>>>(d); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<V1> o(2 * NCHK); CHECK(hipMemcpy(o.data(), d, 2 * NCHK * sizeof(V1), hipMemcpyDeviceToHost)); int bad = 0;
    for (int i = 0; i < NCHK; i++) for (int j = 0; j < 4; j++) bad += o[2*i].d[j] != exp[i].d[j];
    printf("V1 4x64 int128   : %8.3f ms  %8.2f Gmul/s  mismatches %d\n", ms, nmul / ms / 1e6, bad); hipFree(d); }
  // V2
  { V2* d; CHECK(hipMalloc(&d, 2 * nt * sizeof(V2))); CHECK(hipMemcpy(d, h1.data(), 2 * nt * sizeof(V2), hipMemcpyHostToDevice));
    k_v2<<<cus * wps, 256>>>(d); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(d, h1.data(), 2 * nt * sizeof(V2), hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0)); k_v2<<<cus * wps, 256>>>(d); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<V1> o(2 * NCHK); CHECK(hipMemcpy(o.data(), d, 2 * NCHK * sizeof(V1), hipMemcpyDeviceToHost)); int bad = 0;
    for (int i = 0; i < NCHK; i++) for (int j = 0; j < 4; j++) bad += o[2*i].d[j] != exp[i].d[j];
    printf("V2 8x32 CIOS     : %8.3f ms  %8.2f Gmul/s  mismatches %d\n", ms, nmul / ms / 1e6, bad); hipFree(d); }
  // V3: inputs x -> x*2^5 mod p in 9x29 (Montgomery R=2^261 = 2^256 * 2^5); chain result x' relates the same way:
  // mont261(a*2^5, b*2^5) = a*b*2^10/2^261 = (a*b/2^256) * 2^5  -> result = V1 result * 2^5 mod p (mod p; V3 output may be unreduced)
  { std::vector<V3> h3(2 * nt);
    for (size_t i = 0; i < 2 * nt; i++) { uint64_t s4[4]; if (i < 2 * NCHK) shlmod(h1[i].d, 5, s4); else { for (int j = 0; j < 4; j++) s4[j] = h1[i].d[j]; s4[3] &= 0x0fffffffffffffffULL; } uint64_t s5[5] = {s4[0],s4[1],s4[2],s4[3],0}; to29(s5, h3[i].d); }
    V3* d; CHECK(hipMalloc(&d, 2 * nt * sizeof(V3))); CHECK(hipMemcpy(d, h3.data(), 2 * nt * sizeof(V3), hipMemcpyHostToDevice));
    k_v3<<<cus * wps, 256>>>(d); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(d, h3.data(), 2 * nt * sizeof(V3), hipMemcpyHostToDevice));
    CHECK(hipEventRecord(e0)); k_v3<<<cus * wps, 256>>>(d); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<V3> o(2 * NCHK); CHECK(hipMemcpy(o.data(), d, 2 * NCHK * sizeof(V3), hipMemcpyDeviceToHost)); int bad = 0;
    for (int i = 0; i < NCHK; i++) { uint64_t v[5]; from29(o[2*i].d, v); uint64_t p5[5] = {hP64[0],hP64[1],hP64[2],hP64[3],0}; while (ge(v, p5)) sub5(v, p5);
      uint64_t e4[4]; shlmod(exp[i].d, 5, e4); for (int j = 0; j < 4; j++) bad += v[j] != e4[j]; }
    printf("V3 9x29 lazy     : %8.3f ms  %8.2f Gmul/s  mismatches %d\n", ms, nmul / ms / 1e6, bad); hipFree(d); }
  return 0;
}
