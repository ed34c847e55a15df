// msm.hip -- Pippenger multi-scalar multiplication over BN254 G1 for gfx950.
//
// Replaces scalar_multiplication::pippenger / pippenger_internal / compute_wnaf_state
// (reference src/barretenberg/curves/bn254/scalar_multiplication.cpp:265-308,457-476,576-648).
// The reference is a serial bucket method: per scalar an endomorphism split + wNAF (a6,a7), then for each of 8 rounds
// 2n dependent mixed-adds into 2^15 Jacobian buckets with software prefetch, then a serial running sum.  None of
// that shape survives here; what survives is the mathematics (sum_i k_i P_i) and the output contract.
//
// MI355X design (DESIGN.md has the measurements behind each choice):
//   * scalars: from-Montgomery and signed base-2^c digits in one pass (K0).  No GLV split: with the bucket work
//     spread over W * 2^(c-1) independent accumulators the endomorphism buys nothing on a GPU, and reading only the
//     even (base) entries of the caller's endo table halves point traffic.
//   * binning instead of scatter-add: a two-pass counting sort of (window, point) pairs by bucket, histograms and cursors
//     private to a workgroup in LDS, eight digits per 16-byte load, both scatters staged through LDS so that the writes
//     leave as coalesced runs; no global atomics (K1-K3).
//   * accumulation (K4, ~90% of the time): one lane per (window, bucket), XYZZ accumulator in VGPRs, points gathered
//     from the resident SRS in the kernels' Montgomery-261 form; 524,288 independent chains at n = 2^20 keep every
//     SIMD issuing v_mad_u64_u32.  Integer-VALU bound: 2^20 * 16 madds * 10 field multiplies.
//   * bucket reduction (K5): sum_b (b+1) B_b per bucket set without a serial running sum: row sums and column sums of the
//     bucket matrix by log-depth trees, then bit-sliced sums, every point spread over the four lanes of a quad
//     (g1_quad.hpp: four multiplication steps per addition instead of fourteen); the O(c) leftover points per bucket set
//     are combined by the host (host_g1.hpp) and normalised once.
//   * multi-GPU: windows are independent, so rank g takes a window range and returns a normalised partial sum.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "bbgpu_internal.h"
#include "g1.hpp"
#include "g1_quad.hpp"
#include "host_g1.hpp"

namespace bbgpu {

using Fr = FrP;
// Wave priority of the digit / sort kernels.  In the two-deep pipeline they run beside the PREVIOUS MSM's accumulation, stretched from
// 0.23 ms to ~1.07 ms, and end a few tens of microseconds after it -- the next accumulation waits for them.  Priority 2 (above the
// accumulation's 0, below the tail kernels' 3) lets them finish in time: measured 1.400 -> 1.382 ms/step at 2^20 and 0.318 -> 0.305 ms
// for the 2-window share of an 8-way split (tools/msm_ab.py, A/B in one box; priority 3 is no better).
#ifndef BBGPU_FRONT_PRIO
#define BBGPU_FRONT_PRIO 2
#endif
#define FRONT_PRIO() __builtin_amdgcn_s_setprio(BBGPU_FRONT_PRIO)
#ifndef BBGPU_TAIL_PRIO
#define BBGPU_TAIL_PRIO 3 // the tail kernels' wave priority (short dependent chains); A/B builds: -DBBGPU_TAIL_PRIO=0 .. 3
#endif
constexpr int SCALAR_BITS = 254; // r < 2^254 (fr.hpp:12-15)
constexpr int MSM_MAX_C = 16;    // largest window without tables (one bucket set per window); digits stored as int16
// (with tables, one shared bucket set: up to 17-bit windows -> 15 of them at 2^20, digits stored as uint16 magnitude + sign bit; capi.hip picks the width)
constexpr int MSM_THREADS = 256;

__device__ __forceinline__ void ld8(const uint32_t* p, uint32_t (&w)[8])
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void ld16(const uint32_t* p, uint32_t (&w)[16])
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 a = q[i];
        w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w;
    }
}
__device__ __forceinline__ void st32(uint32_t* p, const uint32_t (&w)[32])
{
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void ld32(const uint32_t* p, uint32_t (&w)[32])
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint4 a = q[i];
        w[4 * i] = a.x; w[4 * i + 1] = a.y; w[4 * i + 2] = a.z; w[4 * i + 3] = a.w;
    }
}

// ---- digit windows ---------------------------------------------------------------------------------------------------
struct WinLayout {
    uint16_t off[66]; // off[w] = first bit of window w; off[W] = end
};
static WinLayout make_layout(int c, bool balanced)
{
    WinLayout LO{};
    const int W = msm_num_windows(c);
    if (!balanced) {
        for (int w = 0; w <= W; w++) LO.off[w] = (uint16_t)(c * w);
        return LO;
    }
    // W windows of base or base + 1 bits covering exactly the 254 bits of a canonical scalar, wider ones at the bottom;
    // base = floor(254 / W) <= c - 1, so the (unsigned) top window value <= 2^(c-1) still indexes a bucket
    const int base = SCALAR_BITS / W, rem = SCALAR_BITS - base * W;
    int bit = 0;
    for (int w = 0; w < W; w++) {
        LO.off[w] = (uint16_t)bit;
        bit += base + (w < rem ? 1 : 0);
    }
    LO.off[W] = (uint16_t)bit;
    return LO;
}
// ---------------------------------------------------------------------------------------------------------------------
// SRS: reference endo table (2n x 64 B, Montgomery 2^256) -> resident base points (n x 64 B, Montgomery 2^261, canonical)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void srs_convert_kernel(const uint32_t* __restrict__ table, uint32_t* __restrict__ srs, uint32_t n, uint32_t stride_words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[16];
    ld16(table + (size_t)i * stride_words, w);
    AffineV<2> a;
    load_affine_m256(a, w);
    uint32_t o[16];
    store_affine_m261(o, a.x, a.y);
    uint4* q = reinterpret_cast<uint4*>(srs + (size_t)i * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// resident points -> reference endo table entries 2i (P) and 2i+1 (beta*x, -y), Montgomery 2^256  (a9)
__global__ void srs_export_kernel(const uint32_t* __restrict__ srs, uint32_t* __restrict__ table, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[16];
    ld16(srs + (size_t)i * 16, w);
    AffineV<1> a;
    load_affine_m261(a, w);
    uint32_t o[8];
    uint32_t* dst = table + (size_t)i * 32;
    to_canonical(m261_to_m256<Fq>(a.x), o);
    for (int k = 0; k < 8; k++) dst[k] = o[k];
    to_canonical(m261_to_m256<Fq>(a.y), o);
    for (int k = 0; k < 8; k++) dst[8 + k] = o[k];
    auto bx = mul(a.x, fe_from<Fq>(Fq::BETA));
    to_canonical(m261_to_m256<Fq>(bx), o);
    for (int k = 0; k < 8; k++) dst[16 + k] = o[k];
    to_canonical(m261_to_m256<Fq>(weak(neg(a.y))), o);
    for (int k = 0; k < 8; k++) dst[24 + k] = o[k];
}

// ---- pre-shifted window tables: tab[w * n + i] = 2^off[w] * P_i, affine canonical Montgomery-261 ----------------------
// (the reference's generate_pippenger_precompute_table idea, scalar_multiplication.cpp:90-129, built on the device).
// One lane per base point: off[w] - off[w-1] doublings per window, one inversion per stored point.  Once per SRS.

// Only windows [w_begin, w_end) are stored (and inverted): a rank of an N-way split keeps 1/N of the table; `tab` is the address window 0 WOULD
// have (the allocation starts at window w_begin), so every consumer indexes tab[w * n + i] unchanged.
__global__ void __launch_bounds__(MSM_THREADS) srs_table_kernel(const uint32_t* __restrict__ srs, uint32_t* __restrict__ tab, uint32_t n, WinLayout LO,
                                                              uint32_t num_windows, uint32_t w_begin, uint32_t w_end)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w16[16];
    ld16(srs + (size_t)i * 16, w16);
    if (w_begin == 0) {
        uint4* q = reinterpret_cast<uint4*>(tab + (size_t)i * 16);
#pragma unroll
        for (int t = 0; t < 4; t++) q[t] = make_uint4(w16[4 * t], w16[4 * t + 1], w16[4 * t + 2], w16[4 * t + 3]);
    }
    AffineV<1> p;
    load_affine_m261(p, w16);
    Xyzz acc;
    from_affine(acc, p);
    for (uint32_t w = 1; w < num_windows && w < w_end; w++) {
        for (uint32_t k = LO.off[w - 1]; k < LO.off[w]; k++) {
            Xyzz t;
            dbl(t, acc);
            acc = t;
        }
        if (w < w_begin) continue;
        // 2^(c w) P is never infinity: the group order is an odd prime
        const uint64_t e[4] = { Fq::P64[0] - 2, Fq::P64[1], Fq::P64[2], Fq::P64[3] };
        auto inv = pow_u256<Fq>(mul(acc.zz, acc.zzz), e);
        auto izz = mul(inv, acc.zzz), izzz = mul(inv, acc.zz);
        uint32_t o[16];
        store_affine_m261(o, mul(acc.x, izz), mul(acc.y, izzz));
        uint4* q = reinterpret_cast<uint4*>(tab + ((size_t)w * n + i) * 16);
#pragma unroll
        for (int t = 0; t < 4; t++) q[t] = make_uint4(o[4 * t], o[4 * t + 1], o[4 * t + 2], o[4 * t + 3]);
    }
}

// ---- synthetic SRS x^i G (fixed-base, 8-bit windows) ----------------------------------------------------------------
// tab[w][d] = d * 2^(8w) * G as XYZZ words, d in [0,256) (d = 0: infinity)
__global__ void srs_gen_table_kernel(uint32_t* __restrict__ tab)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= 32) return;
    Xyzz base;
    AffineV<1> g;
    g.x = fe_from<Fq>(Fq::GEN_X);
    g.y = fe_from<Fq>(Fq::GEN_Y);
    from_affine(base, g);
    for (uint32_t k = 0; k < 8 * w; k++) {
        Xyzz t;
        dbl(t, base);
        base = t;
    }
    Xyzz acc;
    set_infinity(acc);
    for (uint32_t d = 0; d < 256; d++) {
        uint32_t o[32];
        store_xyzz(o, acc);
        st32(tab + ((size_t)w * 256 + d) * 32, o);
        Xyzz t;
        add(t, acc, base);
        acc = t;
    }
}
__device__ __forceinline__ Fe<Fq, 1, 2> fq_inverse(const Fe<Fq, 1, 2>& a)
{
    const uint64_t e[4] = { Fq::P64[0] - 2, Fq::P64[1], Fq::P64[2], Fq::P64[3] };
    return pow_u256<Fq>(a, e);
}
// srs[i] = x^(first + i) * G, affine canonical Montgomery-261
__global__ void srs_gen_points_kernel(const uint32_t* __restrict__ tab, Limbs9 x261, uint32_t* __restrict__ srs, uint32_t n, uint32_t first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // s = x^(first + i) in Fr (Montgomery-261), then plain canonical
    Fe<Fr, 1, 2> s = fe_one<Fr>(), b = fe_from<Fr>(x261.d);
    for (uint32_t e = first + i; e; e >>= 1) {
        if (e & 1) s = mul(s, b);
        b = sqr(b);
    }
    uint32_t k[8];
    {
        FeT<Fr> one_raw = fe_zero<Fr>();
        one_raw.d[0] = 1;
        to_canonical(mul(s, one_raw), k); // s * R^-1 = plain value
    }
    Xyzz acc;
    set_infinity(acc);
    for (uint32_t w = 0; w < 32; w++) {
        const uint32_t d = (k[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (d) {
            uint32_t pw[32];
            ld32(tab + ((size_t)w * 256 + d) * 32, pw);
            Xyzz q;
            load_xyzz(q, pw);
            Xyzz t;
            add(t, acc, q);
            acc = t;
        }
    }
    // affine: x = X / ZZ, y = Y / ZZZ (x^i != 0 mod r so the point is finite)
    auto inv = fq_inverse(mul(acc.zz, acc.zzz));
    auto izz = mul(inv, acc.zzz), izzz = mul(inv, acc.zz);
    uint32_t o[16];
    store_affine_m261(o, mul(acc.x, izz), mul(acc.y, izzz));
    uint4* q = reinterpret_cast<uint4*>(srs + (size_t)i * 16);
#pragma unroll
    for (int t = 0; t < 4; t++) q[t] = make_uint4(o[4 * t], o[4 * t + 1], o[4 * t + 2], o[4 * t + 3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// K0: scalars (Montgomery 2^256, any representative) -> signed digits, window-major int16
//     window w covers bits [off[w], off[w+1]) of the canonical scalar; d_w in [-2^(s_w - 1), 2^(s_w - 1)) with a carry into
//     the next window, the top window keeps its value (no carry out);  k = sum_w d_w 2^off[w].
//     Replaces a5 + a6 + a7 (from_montgomery, endo split, wNAF).
//     Two layouts (make_layout): uniform windows of c bits when every window owns a bucket set (the host then combines the
//     windows by c doublings each), and BALANCED windows of c or c - 1 bits when pre-shifted tables feed one shared bucket set:
//     254 = 21 * 12 + 2 would leave a 2-bit top window whose n entries all land in 3 buckets (measured at n = 2^16: one
//     workgroup of the sort and the heavy-bucket merge became the critical path, 130 us of a 570 us MSM).
// ---------------------------------------------------------------------------------------------------------------------
struct ScalarSets {
    const uint32_t* p[MSM_MAX_JOBS]; // one scalar vector per job of a batch (blockIdx.y)
};
// Digit storage: int16 for windows of up to 16 bits.  17-bit windows give digits in [-2^16, 2^16): stored as a uint16 magnitude
// (d >= 0: d, d < 0: -d - 1) plus one sign bit per digit, packed 64 to a word by a wave ballot -- 2.03 bytes per digit instead of the
// 4 of an int32 array, which the histogram and the scatter pass both read in full.
template <class DT> struct DigitTraits { static constexpr bool wide = false; };
template <> struct DigitTraits<uint16_t> { static constexpr bool wide = true; };
template <class DT> __device__ __forceinline__ int load_digit(const DT* __restrict__ dg, const unsigned long long* __restrict__ sg, uint32_t i)
{
    if constexpr (DigitTraits<DT>::wide) {
        const int u = dg[i];
        return ((sg[i >> 6] >> (i & 63u)) & 1ull) ? -u - 1 : u;
    } else {
        return dg[i];
    }
}
// eight consecutive digits of one window, first index a multiple of 8 (one 16-byte load + one byte of sign bits): the sort passes are
// chains of load -> LDS atomic -> store per lane, i.e. bound by the number of dependent global loads a lane issues
template <class DT> __device__ __forceinline__ void load_digits8(const DT* __restrict__ dg, const unsigned long long* __restrict__ sg, uint32_t i8, int (&d)[8])
{
    const uint4 q = *reinterpret_cast<const uint4*>(dg + i8);
    const uint32_t w[4] = { q.x, q.y, q.z, q.w };
    if constexpr (DigitTraits<DT>::wide) {
        const uint32_t sb = reinterpret_cast<const uint8_t*>(sg)[i8 >> 3];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int u = (int)((w[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            d[k] = ((sb >> k) & 1u) ? -u - 1 : u;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = (int)(int16_t)(w[k >> 1] >> (16 * (k & 1)));
    }
}
template <class DT> __global__ void __launch_bounds__(MSM_THREADS) msm_digits_kernel(ScalarSets sets, DT* __restrict__ digits_all, unsigned long long* __restrict__ signs_all,
                                                               uint32_t n, WinLayout LO, uint32_t num_windows, uint32_t wb, uint32_t we)
{
    FRONT_PRIO();
    // only windows [wb, we) are stored (a rank of a window-sharded MSM needs its share only); the carry chain still starts at window 0
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* __restrict__ scalars = sets.p[blockIdx.y];
    DT* __restrict__ digits = digits_all + (size_t)blockIdx.y * num_windows * n;
    const uint32_t s64 = (n + 63) >> 6; // sign words per window (a wave covers 64 consecutive scalars: blockDim is a multiple of 64)
    unsigned long long* __restrict__ signs = signs_all + (size_t)blockIdx.y * num_windows * s64;
    uint32_t w[8], k[9];
    ld8(scalars + (size_t)i * 8, w);
    to_canonical(exact_limbs(mul(unpack<Fr>(w), fe_from<Fr>(Fr::M256_TO_PLAIN))), w); // x*2^256 * 2^5 / 2^261 = x, canonical (a product: no carry chain before packing)
#pragma unroll
    for (int j = 0; j < 8; j++) k[j] = w[j];
    k[8] = 0;
    uint32_t carry = 0;
    for (uint32_t win = 0; win < we; win++) {
        const uint32_t bit = LO.off[win], sz = LO.off[win + 1] - bit, j = bit >> 5, s = bit & 31;
        const uint32_t half = 1u << (sz - 1), mask = (1u << sz) - 1;
        uint32_t v = 0;
        if (j < 8) {
            v = k[j] >> s;
            if (s + sz > 32) v |= k[j + 1] << (32 - s);
        }
        v = (v & mask) + carry;
        carry = (win + 1 < num_windows && v >= half) ? 1u : 0u;
        const int32_t d = (int32_t)v - (int32_t)(carry << sz);
        if (win >= wb) {
            if constexpr (DigitTraits<DT>::wide) {
                const unsigned long long neg = __ballot(d < 0);
                digits[(size_t)win * n + i] = (DT)(d < 0 ? -d - 1 : d);
                if ((threadIdx.x & 63u) == 0) signs[(size_t)win * s64 + (i >> 6)] = neg;
            } else {
                digits[(size_t)win * n + i] = (DT)d;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K1-K3: two-pass counting sort of (window, bucket) -> point index, every pass coalesced.
//   bucket = bin * 2^LB + lo.  Pass A bins entries by (window, bin) -- 2^HB <= 256 bins, histogram and cursors in LDS --
//   so that one workgroup's output per bin is a contiguous run; pass B gives every (window, bin) to one workgroup which
//   counting-sorts its few thousand entries by `lo` inside a 16 KiB neighbourhood that stays in L2.
//   (A single-pass scatter of 4-byte entries into the 64 MiB list was measured at 8x write amplification: PMC
//   WRITE_SIZE 525 MB for 64 MiB of payload, 0.22 ms.)
//   tmp entry: point index (24 bits) | lo << 24 (7 bits) | sign << 31.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 1024;
constexpr uint32_t SORT_MAX_LB = 7;
constexpr uint32_t SORTB_MAX_PARTS = 8;       // pieces a HEAVY bin is cut into by pass B of a large MSM
constexpr uint32_t SORTB_SPLIT_MIN = 1u << 16; // ... a bin with more entries than this

// LDS counter increments for keys that may be heavily duplicated inside a wave.  With uniform digits the 64 lanes of a wave hit ~60 different
// counters and the plain LDS atomic is the right tool; with skewed scalars (every scalar equal, 0 / 1 / -1 witnesses: what real circuits hold) all
// lanes hit ONE counter, the LDS serialises them, and the sort of a 2^20-point MSM went from 0.13 to 2.2 ms (tools/skewed_stages.py).
// wave_keys_heavy(): one test per lane trip (8 entries), on the trip's first entry -- do at least a quarter of the lanes that have one share
// the first such lane's key?  (Wave-uniform among the lanes that call it.)  Only then lds_inc_dedup() peels up to four distinct keys
// per entry slot -- one atomic of the group's size by its first lane, the others take consecutive ranks -- and lets whatever is left fall back
// to the plain atomic.  lds_inc_dedup is called by the lanes that have an entry (divergent code: its ballots see the active lanes only) and
// returns the counter value before this lane's increment, like atomicAdd.
__device__ __forceinline__ bool wave_keys_heavy(bool valid, uint32_t key)
{
    const uint64_t vm = __ballot(valid);
    if (!vm) return false;
    const uint32_t k0 = __builtin_amdgcn_readlane(key, __ffsll((unsigned long long)vm) - 1);
    return 4 * __popcll(__ballot(valid && key == k0)) >= __popcll(vm);
}
__device__ __forceinline__ uint32_t lds_inc_dedup(uint32_t* ctr, uint32_t key, bool heavy)
{
    if (!heavy) return atomicAdd(&ctr[key], 1u); // wave-uniform
    const uint32_t lane = __lane_id();
    uint64_t todo = __ballot(1);
    uint32_t res = 0;
    for (int it = 0; it < 4 && todo; it++) { // wave-uniform loop
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t kk = __builtin_amdgcn_readlane(key, leader);
        const uint64_t m = __ballot(key == kk) & todo;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&ctr[kk], (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readlane(base, leader);
        if ((m >> lane) & 1ull) res = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
    }
    if ((todo >> lane) & 1ull) res = atomicAdd(&ctr[key], 1u);
    return res;
}

// A "group" is a set of `wpg` consecutive windows that share one bucket set: 1 window per group normally, all windows of
// the call in one group when the SRS carries pre-shifted window tables (the window weight is then baked into the point).
template <class DT> __global__ void __launch_bounds__(SORT_THREADS) __attribute__((amdgpu_num_vgpr(32))) sortA_hist_kernel(const DT* __restrict__ digits, const unsigned long long* __restrict__ signs, uint32_t* __restrict__ histA, uint32_t n,
                                                                uint32_t bins, uint32_t lb, uint32_t slices, uint32_t slice_len, uint32_t win0,
                                                                uint32_t wpg, uint32_t first_i0, uint32_t last_i1, uint32_t blo, uint32_t bcnt)
{
    // blo, bcnt: a bucket-range share (msm_issue_buckets) keeps only the digits whose bucket lies in [blo, blo + bcnt); all three
    // pass-A kernels apply the same test, everything after them sees a list that simply has no entries elsewhere
    FRONT_PRIO();
    __shared__ uint32_t lh[SORT_THREADS];
    const uint32_t s = blockIdx.x, wl = blockIdx.y;
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = s * slice_len, hi = min(n, lo + slice_len);
    const uint32_t s64 = (n + 63) >> 6;
    if ((n & 7u) == 0) { // rows are 16-byte aligned: 8 digits per load, the (window, block of 8) pairs of the slice dealt round-robin to the lanes
        const uint32_t lo8 = lo & ~7u, bpw = lo < hi ? (((hi + 7u) & ~7u) - lo8) >> 3 : 0u;
        for (uint32_t u = threadIdx.x; u < wpg * bpw; u += SORT_THREADS) {
            const uint32_t k = u / bpw, i8 = lo8 + (u - k * bpw) * 8;
            const uint32_t klo = (k == 0) ? max(lo, first_i0) : lo, khi = (k + 1 == wpg) ? min(hi, last_i1) : hi;
            int d[8];
            load_digits8<DT>(digits + (size_t)(win0 + wl * wpg + k) * n, signs + (size_t)(win0 + wl * wpg + k) * s64, i8, d);
            const bool heavy = wave_keys_heavy(d[0] != 0, (uint32_t)((d[0] < 0 ? -d[0] : d[0]) - 1) >> lb); // among the lanes still in the loop
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (d[j] && i8 + j >= klo && i8 + j < khi) {
                    const uint32_t b = (uint32_t)((d[j] < 0 ? -d[j] : d[j]) - 1);
                    if (b - blo < bcnt) (void)lds_inc_dedup(lh, b >> lb, heavy);
                }
        }
    } else {
        for (uint32_t k = 0; k < wpg; k++) {
            const DT* dg = digits + (size_t)(win0 + wl * wpg + k) * n;
            const unsigned long long* sg = signs + (size_t)(win0 + wl * wpg + k) * s64;
            // a row-range share (msm_issue_rows) owns only points >= first_i0 of its first window and < last_i1 of its last one
            const uint32_t klo = (k == 0) ? max(lo, first_i0) : lo, khi = (k + 1 == wpg) ? min(hi, last_i1) : hi;
            for (uint32_t i = klo + threadIdx.x; i < khi; i += SORT_THREADS) {
                const int d = load_digit<DT>(dg, sg, i);
                if (d) {
                    const uint32_t b = (uint32_t)((d < 0 ? -d : d) - 1);
                    if (b - blo < bcnt) atomicAdd(&lh[b >> lb], 1u);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < bins) histA[((size_t)wl * slices + s) * bins + threadIdx.x] = lh[threadIdx.x];
}

// Column scan of the pass-A histogram matrix [slices][bins] of one group: for every bin the exclusive prefix over slices
// (cursors relative to the bin start, in place) and the bin total.  16 bins per block, one wave per bin, each lane owns
// a run of slices and the wave combines them with shuffles (a single thread walking hundreds of slices was latency-bound).
__global__ void __launch_bounds__(SORT_THREADS) sortA_colscan_kernel(uint32_t* __restrict__ histA, uint32_t* __restrict__ bintot, uint32_t bins,
                                                                   uint32_t slices)
{
    FRONT_PRIO();
    const uint32_t wl = blockIdx.y, lane = threadIdx.x & 63;
    const uint32_t bin = blockIdx.x * (SORT_THREADS / 64) + (threadIdx.x >> 6);
    if (bin >= bins) return; // whole waves exit together
    uint32_t* H = histA + (size_t)wl * slices * bins + bin;
    const uint32_t spp = (slices + 63) / 64;
    const uint32_t s0 = min(slices, lane * spp), s1 = min(slices, s0 + spp);
    uint32_t sum = 0;
    for (uint32_t s = s0; s < s1; s++) sum += H[(size_t)s * bins];
    uint32_t incl = sum; // inclusive scan over the 64 lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    uint32_t run = incl - sum;
    for (uint32_t s = s0; s < s1; s++) {
        const uint32_t cnt = H[(size_t)s * bins];
        H[(size_t)s * bins] = run;
        run += cnt;
    }
    if (lane == 63) bintot[(size_t)wl * bins + bin] = incl;
}
// per group: exclusive scan of the bin totals -> local bin starts, and the group's entry count (bins <= 1024)
// solo_bases (a launch with ONE group -- a single MSM against window tables, the common case): there is nothing to add up across groups, so this workgroup
// writes the bases and the list's end itself and sort_bases_kernel is not launched (round 5: one dependent launch, ~5 us of a small MSM's chain).
__global__ void __launch_bounds__(SORT_THREADS) sortA_scan_kernel(const uint32_t* __restrict__ bintot, uint32_t* __restrict__ binstart,
                                                                uint32_t* __restrict__ totals, uint32_t bins, uint32_t* __restrict__ solo_bases,
                                                                uint32_t* __restrict__ solo_gstart_end)
{
    FRONT_PRIO();
    __shared__ uint32_t part[SORT_THREADS];
    const uint32_t wl = blockIdx.x, t = threadIdx.x;
    const uint32_t mine = t < bins ? bintot[(size_t)wl * bins + t] : 0;
    part[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < SORT_THREADS; off <<= 1) {
        uint32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (t < bins) binstart[(size_t)wl * bins + t] = part[t] - mine;
    if (t == SORT_THREADS - 1) {
        totals[wl] = part[SORT_THREADS - 1];
        if (solo_bases) { // gridDim.x == 1
            solo_bases[0] = 0;
            solo_bases[1] = part[SORT_THREADS - 1];
            solo_gstart_end[0] = part[SORT_THREADS - 1];
            solo_gstart_end[1] = 0xffffffffu; // sentinel, as in sort_bases_kernel
        }
    }
}
// window bases (exclusive prefix of the window totals), M = total number of entries -> gstart[total_buckets]
__global__ void sort_bases_kernel(const uint32_t* __restrict__ totals, uint32_t* __restrict__ bases, uint32_t* __restrict__ gstart_end, uint32_t nw)
{
    FRONT_PRIO();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < nw; w++) {
            bases[w] = run;
            run += totals[w];
        }
        bases[nw] = run;
        gstart_end[0] = run;
        gstart_end[1] = 0xffffffffu; // sentinel: the accumulation's last lane walks past the last bucket into a dummy one (msm_accumulate_kernel)
    }
}

// (Round 5 tried these three kernels as ONE launch -- the two after the column scan are nothing but launch latency, 4.8 + 5.1 us of a 253 us MSM of 2^16
// points -- twice: the whole matrix of a group on one workgroup, and the column scan as above with the LAST workgroup to arrive scanning the bin totals and
// laying down the bases.  Both were SLOWER than the three launches in one-box A/Bs, 0.277 against 0.257-0.261 ms per 2^16-point MSM (profiles/r05_scan_fused_ab.txt):
// the agent-scope release / acquire around the arrival counter writes back and invalidates the XCD's L2, in every workgroup, where a kernel boundary does it once.)

template <class DT> __global__ void __launch_bounds__(SORT_THREADS) __attribute__((amdgpu_num_vgpr(32))) sortA_scatter_kernel(const DT* __restrict__ digits, const unsigned long long* __restrict__ signs, const uint32_t* __restrict__ cursorsA,
                                                                   const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ bases,
                                                                   uint32_t* __restrict__ tmp, uint32_t n, uint32_t bins, uint32_t lb,
                                                                   uint32_t slices, uint32_t slice_len, uint32_t win0, uint32_t wpg,
                                                                   uint32_t idx_stride, uint32_t windows_per_job, uint32_t first_i0, uint32_t last_i1, uint32_t blo, uint32_t bcnt)
{
    FRONT_PRIO();
    __shared__ uint32_t lc[SORT_THREADS];
    const uint32_t s = blockIdx.x, wl = blockIdx.y;
    if (threadIdx.x < bins)
        lc[threadIdx.x] = cursorsA[((size_t)wl * slices + s) * bins + threadIdx.x] + binstart[(size_t)wl * bins + threadIdx.x] + bases[wl];
    __syncthreads();
    const uint32_t lo = s * slice_len, hi = min(n, lo + slice_len);
    const uint32_t lomask = (1u << lb) - 1;
    const uint32_t s64 = (n + 63) >> 6;
    if ((n & 7u) == 0) { // as in sortA_hist_kernel (the two kernels must visit exactly the same entries; their order inside a bin is free)
        const uint32_t lo8 = lo & ~7u, bpw = lo < hi ? (((hi + 7u) & ~7u) - lo8) >> 3 : 0u;
        for (uint32_t u = threadIdx.x; u < wpg * bpw; u += SORT_THREADS) {
            const uint32_t k = u / bpw, i8 = lo8 + (u - k * bpw) * 8;
            const uint32_t wabs = win0 + wl * wpg + k;
            const uint32_t row = (wabs % windows_per_job) * idx_stride;
            const uint32_t klo = (k == 0) ? max(lo, first_i0) : lo, khi = (k + 1 == wpg) ? min(hi, last_i1) : hi;
            int d[8];
            load_digits8<DT>(digits + (size_t)wabs * n, signs + (size_t)wabs * s64, i8, d);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (d[j] && i8 + j >= klo && i8 + j < khi) {
                    const uint32_t b = (uint32_t)((d[j] < 0 ? -d[j] : d[j]) - 1);
                    if (b - blo < bcnt) {
                        const uint32_t pos = atomicAdd(&lc[b >> lb], 1u);
                        tmp[pos] = (row + i8 + j) | ((b & lomask) << 24) | (d[j] < 0 ? 0x80000000u : 0u);
                    }
                }
            }
        }
        return;
    }
    for (uint32_t k = 0; k < wpg; k++) {
        const uint32_t wabs = win0 + wl * wpg + k;
        const DT* dg = digits + (size_t)wabs * n;
        const unsigned long long* sg = signs + (size_t)wabs * s64;
        const uint32_t row = (wabs % windows_per_job) * idx_stride; // row of the pre-shifted table (0 without tables); batches repeat the windows per job
        const uint32_t klo = (k == 0) ? max(lo, first_i0) : lo, khi = (k + 1 == wpg) ? min(hi, last_i1) : hi;
        for (uint32_t i = klo + threadIdx.x; i < khi; i += SORT_THREADS) {
            const int d = load_digit<DT>(dg, sg, i);
            if (d) {
                const uint32_t b = (uint32_t)((d < 0 ? -d : d) - 1);
                if (b - blo < bcnt) {
                    const uint32_t pos = atomicAdd(&lc[b >> lb], 1u);
                    tmp[pos] = (row + i) | ((b & lomask) << 24) | (d < 0 ? 0x80000000u : 0u);
                }
            }
        }
    }
}

// Pass A with the scatter staged through LDS (see sortB_staged_kernel for why): a tile = one 16-byte digit load per lane (<= 8,192 entries)
// is ranked by bin with LDS atomics, the tile histogram is scanned by the workgroup, the entries are laid out bin by bin in LDS (with
// their bin beside them: the entry word has no room for it) and written out with consecutive lanes on consecutive addresses.
// Rows must be 16-byte aligned (n % 8 == 0); visits exactly the entries sortA_hist_kernel counted.
template <class DT> __global__ void __launch_bounds__(SORT_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) sortA_scatter_staged_kernel(const DT* __restrict__ digits, const unsigned long long* __restrict__ signs, const uint32_t* __restrict__ cursorsA,
                                                                          const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ bases,
                                                                          uint32_t* __restrict__ tmp, uint32_t n, uint32_t bins, uint32_t lb,
                                                                          uint32_t slices, uint32_t slice_len, uint32_t win0, uint32_t wpg,
                                                                          uint32_t idx_stride, uint32_t windows_per_job, uint32_t first_i0, uint32_t last_i1, uint32_t blo, uint32_t bcnt)
{
    FRONT_PRIO();
    constexpr uint32_t TILE = SORT_THREADS * 8;
    __shared__ uint32_t lc[SORT_THREADS];   // next global position per bin
    __shared__ uint32_t th[SORT_THREADS];   // tile histogram, then first slot of the bin in the tile buffer
    __shared__ uint32_t dlt[SORT_THREADS];  // global position - slot
    __shared__ uint32_t wsum[SORT_THREADS / 64 + 1];
    __shared__ uint32_t buf[TILE];
    __shared__ uint16_t bbin[TILE];
    const uint32_t s = blockIdx.x, wl = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    lc[t] = t < bins ? cursorsA[((size_t)wl * slices + s) * bins + t] + binstart[(size_t)wl * bins + t] + bases[wl] : 0u;
    const uint32_t lo = s * slice_len, hi = min(n, lo + slice_len);
    const uint32_t lomask = (1u << lb) - 1;
    const uint32_t s64 = (n + 63) >> 6;
    const uint32_t lo8 = lo & ~7u, bpw = lo < hi ? (((hi + 7u) & ~7u) - lo8) >> 3 : 0u;
    const uint32_t total_u = wpg * bpw;
    for (uint32_t u0 = 0; u0 < total_u; u0 += SORT_THREADS) { // workgroup-uniform trip count
        th[t] = 0;
        __syncthreads();
        uint32_t ent[8], bn[8], rk[8];
        const uint32_t u = u0 + t;
#pragma unroll
        for (int j = 0; j < 8; j++) bn[j] = 0xffffffffu;
        if (u < total_u) {
            const uint32_t k = u / bpw, i8 = lo8 + (u - k * bpw) * 8;
            const uint32_t wabs = win0 + wl * wpg + k;
            const uint32_t row = (wabs % windows_per_job) * idx_stride;
            const uint32_t klo = (k == 0) ? max(lo, first_i0) : lo, khi = (k + 1 == wpg) ? min(hi, last_i1) : hi;
            int d[8];
            load_digits8<DT>(digits + (size_t)wabs * n, signs + (size_t)wabs * s64, i8, d);
            const bool heavy = wave_keys_heavy(d[0] != 0, (uint32_t)((d[0] < 0 ? -d[0] : d[0]) - 1) >> lb);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (d[j] && i8 + j >= klo && i8 + j < khi) {
                    const uint32_t b = (uint32_t)((d[j] < 0 ? -d[j] : d[j]) - 1);
                    if (b - blo < bcnt) {
                        bn[j] = b >> lb;
                        ent[j] = (row + i8 + j) | ((b & lomask) << 24) | (d[j] < 0 ? 0x80000000u : 0u);
                        rk[j] = lds_inc_dedup(th, bn[j], heavy);
                    }
                }
            }
        }
        __syncthreads();
        // exclusive scan of the tile histogram over the bins (one value per lane; th is zero beyond `bins`)
        const uint32_t c = th[t];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += wsum[w];
        const uint32_t excl = before + incl - c;
        th[t] = excl;
        const uint32_t g = lc[t];
        dlt[t] = g - excl;
        lc[t] = g + c;
        if (t == SORT_THREADS - 1) wsum[SORT_THREADS / 64] = excl + c; // entries in this tile
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (bn[j] != 0xffffffffu) {
                const uint32_t slot = th[bn[j]] + rk[j];
                buf[slot] = ent[j];
                bbin[slot] = (uint16_t)bn[j];
            }
        }
        __syncthreads();
        const uint32_t tile_n = wsum[SORT_THREADS / 64];
        for (uint32_t x = t; x < tile_n; x += SORT_THREADS) tmp[dlt[bbin[x]] + x] = buf[x];
        // the next trip's barriers order these reads before wsum / buf / dlt are rewritten (all written after its second barrier;
        // th, cleared before its first barrier, is not read here)
    }
}

// pass B: one workgroup per (bin, window): counting sort by `lo`, emits the final entries and the global bucket starts
__global__ void __launch_bounds__(SORT_THREADS) sortB_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binstart,
                                                   const uint32_t* __restrict__ bases, uint32_t* __restrict__ sorted, uint32_t* __restrict__ gstart,
                                                   uint32_t bins, uint32_t lb, uint32_t nb)
{
    FRONT_PRIO();
    __shared__ uint32_t cnt[128];
    __shared__ uint32_t cur[128];
    const uint32_t bin = blockIdx.x, wl = blockIdx.y, t = threadIdx.x;
    const uint32_t nlo = 1u << lb;
    const uint32_t start = bases[wl] + binstart[(size_t)wl * bins + bin];
    const uint32_t end = (bin + 1 < bins) ? bases[wl] + binstart[(size_t)wl * bins + bin + 1] : bases[wl + 1];
    if (t < 128) cnt[t] = 0;
    __syncthreads();
    // eight independent loads per lane and trip (a lane's entries are blockDim apart: every load of the wave is one coalesced run)
    constexpr int UB = 8;
    for (uint32_t e0 = start + t; e0 < end; e0 += UB * blockDim.x) {
        uint32_t v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const uint32_t e = e0 + k * blockDim.x;
            v[k] = e < end ? tmp[e] : 0u;
        }
#pragma unroll
        for (int k = 0; k < UB; k++)
            if (e0 + k * blockDim.x < end) atomicAdd(&cnt[(v[k] >> 24) & 0x7f], 1u);
    }
    __syncthreads();
    if (t < 64) { // exclusive scan of the <= 128 counters by one wave, two counters per lane
        const uint32_t a = cnt[2 * t], b2 = cnt[2 * t + 1], sum = a + b2;
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)t >= off) incl += o;
        }
        cur[2 * t] = start + incl - sum;
        cur[2 * t + 1] = start + incl - sum + a;
    }
    __syncthreads();
    if (t < nlo) gstart[(size_t)wl * nb + (size_t)bin * nlo + t] = cur[t];
    __syncthreads();
    for (uint32_t e0 = start + t; e0 < end; e0 += UB * blockDim.x) {
        uint32_t v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const uint32_t e = e0 + k * blockDim.x;
            v[k] = e < end ? tmp[e] : 0u;
        }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            if (e0 + k * blockDim.x < end) {
                const uint32_t pos = atomicAdd(&cur[(v[k] >> 24) & 0x7f], 1u);
                sorted[pos] = v[k] & 0x80ffffffu;
            }
        }
    }
}

// Pass B with the scatter staged through LDS.  A lane's `sorted[pos] = v` of sortB_kernel is a 4-byte write to one of 128 runs: 64 requests
// per wave store, and the L2 takes ~128 requests per clock chip-wide -- 15.7M entries = 58 us of request issue alone (measured 74 us).
// Here a tile of 8 entries per lane is ranked by sub-bucket in LDS (LDS atomics), laid out sub-bucket by sub-bucket in a 32 KiB
// buffer, and written out with consecutive lanes on consecutive addresses: runs of ~64 entries = two full 128-byte lines per sub-bucket and tile.
// Large MSMs (round 3): a bin is cut into `parts` contiguous pieces, one workgroup each (blockIdx.z) -- under skewed digit distributions a
// single bin holds up to n entries (every scalar equal: 15 bins of 2^20), and one workgroup walking 10^6 entries was ~1 ms of a 1.25 ms
// sort.  The pieces' sub-bucket counts come from sortB_count_kernel (histB[group][bin][part][128]); a piece places its entries behind
// those of the earlier pieces of its bin, so the result is the same stable order.  parts == 1 (small MSMs): the kernel counts its bin itself.
template <int THREADS> __global__ void __launch_bounds__(THREADS) sortB_count_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binstart,
                                                                                       const uint32_t* __restrict__ bases, uint32_t* __restrict__ histB, uint32_t bins, uint32_t parts)
{
    FRONT_PRIO();
    constexpr int UB = 8;
    constexpr uint32_t TILE = THREADS * UB;
    __shared__ uint32_t cnt[128];
    const uint32_t bin = blockIdx.x, wl = blockIdx.y, part = blockIdx.z, t = threadIdx.x;
    const uint32_t bstart = bases[wl] + binstart[(size_t)wl * bins + bin];
    const uint32_t bend = (bin + 1 < bins) ? bases[wl] + binstart[(size_t)wl * bins + bin + 1] : bases[wl + 1];
    const uint32_t len = bend - bstart;
    if (len <= SORTB_SPLIT_MIN) return; // an ordinary bin: its one workgroup of the scatter kernel counts it itself
    const uint32_t start = bstart + (uint32_t)(((uint64_t)len * part) / parts), end = bstart + (uint32_t)(((uint64_t)len * (part + 1)) / parts);
    if (t < 128) cnt[t] = 0;
    __syncthreads();
    for (uint32_t e0 = start + t; e0 < end; e0 += TILE) {
        uint32_t v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const uint32_t e = e0 + k * THREADS;
            v[k] = e < end ? tmp[e] : 0u;
        }
        const bool heavy = wave_keys_heavy(e0 < end, (v[0] >> 24) & 0x7f);
#pragma unroll
        for (int k = 0; k < UB; k++)
            if (e0 + k * THREADS < end) (void)lds_inc_dedup(cnt, (v[k] >> 24) & 0x7f, heavy);
    }
    __syncthreads();
    if (t < 128) histB[(((size_t)wl * bins + bin) * parts + part) * 128 + t] = cnt[t];
}
template <int THREADS> __global__ void __launch_bounds__(THREADS) sortB_staged_kernel(const uint32_t* __restrict__ tmp, const uint32_t* __restrict__ binstart,
                                                                                        const uint32_t* __restrict__ bases, uint32_t* __restrict__ sorted,
                                                                                        uint32_t* __restrict__ gstart, uint32_t bins, uint32_t lb, uint32_t nb,
                                                                                        const uint32_t* __restrict__ histB, uint32_t parts_launched)
{
    FRONT_PRIO();
    constexpr int UB = 8;
    constexpr uint32_t TILE = THREADS * UB;
    __shared__ uint32_t cnt[128];   // entries per sub-bucket: whole bin, then per tile
    __shared__ uint32_t cur[128];   // next global position per sub-bucket
    __shared__ uint32_t toff[128];  // first slot of the sub-bucket in the tile buffer
    __shared__ uint32_t dlt[128];   // global position - slot
    __shared__ uint32_t bef[128];   // entries of the sub-bucket in the earlier pieces of this bin
    __shared__ uint32_t buf[TILE];
    const uint32_t bin = blockIdx.x, wl = blockIdx.y, part = blockIdx.z, t = threadIdx.x;
    const uint32_t nlo = 1u << lb;
    const uint32_t bstart = bases[wl] + binstart[(size_t)wl * bins + bin];
    const uint32_t bend = (bin + 1 < bins) ? bases[wl] + binstart[(size_t)wl * bins + bin + 1] : bases[wl + 1];
    const uint32_t len = bend - bstart;
    // only HEAVY bins (more than SORTB_SPLIT_MIN entries: skewed digits; a bin of uniform 2^20-point digits holds ~30 k) are cut into pieces;
    // an ordinary bin is one workgroup's as before and the other pieces' workgroups leave at once
    const uint32_t parts = (parts_launched > 1 && len > SORTB_SPLIT_MIN) ? parts_launched : 1u;
    if (part >= parts) return;
    // this workgroup's piece of the bin (the whole bin when parts == 1)
    const uint32_t start = bstart + (uint32_t)(((uint64_t)len * part) / parts), end = bstart + (uint32_t)(((uint64_t)len * (part + 1)) / parts);
    if (t < 128) {
        uint32_t total = 0, before = 0;
        if (parts > 1) {
            const uint32_t* h = histB + (((size_t)wl * bins + bin) * parts) * 128 + t;
            for (uint32_t p = 0; p < parts; p++) {
                const uint32_t c = h[(size_t)p * 128];
                total += c;
                if (p < part) before += c;
            }
        }
        cnt[t] = total;
        bef[t] = before;
    }
    __syncthreads();
    for (uint32_t e0 = start + t; parts == 1 && e0 < end; e0 += TILE) {
        uint32_t v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const uint32_t e = e0 + k * THREADS;
            v[k] = e < end ? tmp[e] : 0u;
        }
        const bool heavy = wave_keys_heavy(e0 < end, (v[0] >> 24) & 0x7f);
#pragma unroll
        for (int k = 0; k < UB; k++)
            if (e0 + k * THREADS < end) (void)lds_inc_dedup(cnt, (v[k] >> 24) & 0x7f, heavy);
    }
    __syncthreads();
    if (t < 64) { // exclusive scan of the <= 128 counters by one wave, two counters per lane
        const uint32_t a = cnt[2 * t], b2 = cnt[2 * t + 1], sum = a + b2;
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)t >= off) incl += o;
        }
        cur[2 * t] = bstart + incl - sum;
        cur[2 * t + 1] = bstart + incl - sum + a;
    }
    __syncthreads();
    if (t < nlo && part == 0) gstart[(size_t)wl * nb + (size_t)bin * nlo + t] = cur[t];
    __syncthreads();
    if (t < 128) cur[t] += bef[t]; // this piece writes behind the earlier pieces of its bin
    __syncthreads();
    for (uint32_t base = start; base < end; base += TILE) { // workgroup-uniform trip count
        if (t < 128) cnt[t] = 0;
        __syncthreads();
        uint32_t v[UB], rk[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) {
            const uint32_t e = base + k * THREADS + t;
            v[k] = e < end ? tmp[e] : 0u;
        }
        const bool heavy = wave_keys_heavy(base + t < end, (v[0] >> 24) & 0x7f);
#pragma unroll
        for (int k = 0; k < UB; k++) {
            rk[k] = 0u;
            if (base + k * THREADS + t < end) rk[k] = lds_inc_dedup(cnt, (v[k] >> 24) & 0x7f, heavy);
        }
        __syncthreads();
        if (t < 64) {
            const uint32_t a = cnt[2 * t], b2 = cnt[2 * t + 1], sum = a + b2;
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if ((int)t >= off) incl += o;
            }
            const uint32_t x0 = incl - sum, x1 = x0 + a;
            toff[2 * t] = x0;
            toff[2 * t + 1] = x1;
            const uint32_t c0 = cur[2 * t], c1 = cur[2 * t + 1];
            dlt[2 * t] = c0 - x0;
            dlt[2 * t + 1] = c1 - x1;
            cur[2 * t] = c0 + a;
            cur[2 * t + 1] = c1 + b2;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < UB; k++)
            if (base + k * THREADS + t < end) buf[toff[(v[k] >> 24) & 0x7f] + rk[k]] = v[k];
        __syncthreads();
        const uint32_t tile_n = min(TILE, end - base);
        for (uint32_t x = t; x < tile_n; x += THREADS) {
            const uint32_t w = buf[x];
            sorted[dlt[(w >> 24) & 0x7f] + x] = w & 0x80ffffffu;
        }
        // the next trip's first barrier (after clearing cnt) orders these reads of buf / dlt before they are overwritten:
        // buf and dlt are written only after that trip's second barrier
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K4: bucket accumulation -- the hot loop (replaces scalar_multiplication.cpp:604-617)
//   The sorted entry list (all windows, ordered by bucket) is cut into chunks of exactly `ch` entries, one lane per
//   chunk, so every lane performs the same number of mixed additions whatever the digit distribution (the top window
//   of 252..254-bit scalars touches only a fraction of its buckets, and real witnesses are skewed).  A lane walks its
//   chunk with an XYZZ accumulator in VGPRs and flushes a partial sum whenever the bucket changes; partial (b, t) of
//   bucket b and chunk t lands in slot b + t, which is injective and keeps each bucket's partials contiguous.
//   K4m then adds the 1-3 partials of every bucket.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_raw(uint32_t* dst, const Xyzz& p)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BBGPU_STORE_RAW_X4)
    // 8-byte stores off one base address.  The compiler's own choice -- nine 16-byte stores -- wants quadruples of adjacent registers,
    // and the accumulator's limbs sit wherever the in-place products of the hot loop leave them: it then shuffles 8 limbs out and
    // back on EVERY trip (22 v_mov + 11 v_mov_b64 in the hot path); pairs it manages to keep adjacent (6 copies left, in the latch).
    uint32_t v[4 * NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { v[i] = p.x.d[i]; v[NL + i] = p.y.d[i]; v[2 * NL + i] = p.zz.d[i]; v[3 * NL + i] = p.zzz.d[i]; }
#pragma unroll
    for (int i = 0; i < 4 * NL; i += 2)
        asm volatile("global_store_dwordx2 %0, %1, off offset:%2" :: "v"(dst), "v"(((unsigned long long)v[i + 1] << 32) | v[i]), "n"(4 * i) : "memory");
#else
#pragma unroll
    for (int i = 0; i < NL; i++) {
        dst[i] = p.x.d[i];
        dst[NL + i] = p.y.d[i];
        dst[2 * NL + i] = p.zz.d[i];
        dst[3 * NL + i] = p.zzz.d[i];
    }
#endif
}
__device__ __forceinline__ void load_raw(Xyzz& p, const uint32_t* src)
{
#pragma unroll
    for (int i = 0; i < NL; i++) {
        p.x.d[i] = src[i];
        p.y.d[i] = src[NL + i];
        p.zz.d[i] = src[2 * NL + i];
        p.zzz.d[i] = src[3 * NL + i];
    }
}
// Register budget of the accumulation: with the asm products the allocator settles at 132 VGPRs (three waves per SIMD, which is what the
// LDS reservation admits anyway).  Holding it to 128 like the tail kernels (-DBBGPU_ACC_CAP128: 4 spills, one scratch access in the hot
// loop) was measured in one box at 1.272 vs 1.259 ms per pipelined 2^20 step: no gain, the default stays uncapped.
#ifdef BBGPU_ACC_CAP128
#define ACC_VGPR_CAP __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define ACC_VGPR_CAP
#endif
constexpr int RAW_WORDS = 4 * NL; // 36 words per partial: lazy limbs, no canonicalisation on the hot path
// The heavy-bucket queue of the merge: heavy[0] = number of queued buckets, heavy[1 .. HEAVY_WGS] = per-bucket arrival counters of the
// workgroups that share a bucket (K4h), heavy[HEAVY_IDS ..] = bucket ids; HEAVY_WGS raw partial sums follow the id list (MsmCarve).
constexpr uint32_t HEAVY_WGS = 256;           // workgroups of K4h
constexpr uint32_t HEAVY_IDS = 1 + HEAVY_WGS; // first bucket id
// block 0 of the accumulation zeroes the HEAVY_IDS header words with one lane each (+ one), and the last arriver of K4h loads the K <= HEAVY_WGS
// slice sums one per lane: both need a workgroup as wide as the queue
static_assert(MSM_THREADS == (int)HEAVY_WGS, "heavy-bucket queue header is cleared / combined by one lane per K4h workgroup");

__global__ void __launch_bounds__(MSM_THREADS) ACC_VGPR_CAP msm_accumulate_kernel(const uint32_t* __restrict__ srs, const uint32_t* __restrict__ sorted,
                                                                   const uint32_t* __restrict__ gstart, uint32_t* __restrict__ partials,
                                                                   uint32_t total_buckets, uint32_t ch, uint32_t prio, uint32_t* __restrict__ heavy_counter)
{
    // small MSMs are a chain of dependent launches: the merge's heavy-bucket counter is zeroed here instead of by a fill launch of its own
    if (heavy_counter && blockIdx.x == 0) { // ... and the arrival counters behind it (HEAVY_IDS words; MSM_THREADS == HEAVY_WGS lanes + one)
        heavy_counter[threadIdx.x] = 0;
        if (threadIdx.x == 0) heavy_counter[MSM_THREADS] = 0;
    }
    // wave priority above the memory-bound sort kernels of the NEXT MSM that share the CUs in the two-deep pipeline (they have
    // a whole accumulation of slack), below the latency-bound tail kernels of the previous one (s_setprio 3)
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = gstart[total_buckets];
    const uint32_t p0 = t * ch;
    if (p0 >= M) return;
    const uint32_t p1 = min(M, p0 + ch);
    // bucket containing entry p0: last b with gstart[b] <= p0
    uint32_t lo = 0, hi = total_buckets; // invariant: gstart[lo] <= p0 < gstart[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (gstart[mid] <= p0) lo = mid; else hi = mid;
    }
    uint32_t b = lo;
    uint32_t next_end = gstart[b + 1];
    Xyzz acc;
    set_infinity(acc);
    uint32_t after_next = gstart[min(b + 2, total_buckets + 1)]; // one boundary ahead, so that a flush does not wait for its load
    bool acc_inf = true; // the accumulator is the point at infinity (as a flag): at the start of the chunk and after P + (-P)
    // Software pipeline (round 3 shape).  Three things are in flight while the ~2,100 VALU instructions of one mixed addition run:
    // the 64-byte table row of entry e+1 (gather from HBM: the window tables exceed the Infinity Cache), the sorted-list word of
    // entry e+2 (so that the next gather's address is known when it is issued), and the bucket boundary after the next.  The row
    // is unpacked into the operand registers at the END of the iteration (its load has had the whole addition to land) and the
    // next gather is issued into the registers that frees -- no second row buffer and no register-to-register copy of the row
    // (round 2 copied 16 words per addition); the operand itself dies in the addition's first two products.
    uint32_t v = sorted[p0];
    uint32_t vn = sorted[p0 + 1 < p1 ? p0 + 1 : p0];
    uint32_t w[16];
    ld16(srs + (size_t)(v & 0x7fffffffu) * 16, w);
    Fe<Fq, 1, 1> px;
    Fe<Fq, 1, 2> py;
    load_affine_m261_signed(px, py, w, (v >> 31) != 0);
    ld16(srs + (size_t)(vn & 0x7fffffffu) * 16, w);
    // Every lane runs exactly `ch` trips -- a wave-uniform count: with a per-lane exit the compiler keeps the live-out values of
    // the lanes that left in a second set of registers (38 copies per trip), and a per-lane "still has entries" predicate around
    // the addition costs 40 more registers (190: two waves per SIMD).  Only the ONE lane of the grid that holds the end of the
    // list has fewer entries: at e == M it flushes its last bucket like any other and walks on into a DUMMY bucket (index
    // total_buckets; gstart[total_buckets + 1] is a sentinel no entry reaches, written by sort_bases_kernel), where its remaining
    // trips re-add its last point to a sum nobody reads (slot total_buckets + t lies inside the partials array: MsmCarve keeps one
    // chunk more than any grid has; the merge kernels stop at total_buckets).
    uint32_t e = p0;
    uint32_t trips = ch; // ch >= 1; a scalar count tested at the bottom: the values the final store reads are then the loop's own registers
    do {
        const uint32_t vnn = sorted[min(e + 2, p1 - 1)];
        // One-sided branches only (see madd_ip): a lane either starts a bucket with this entry -- after flushing the finished
        // bucket's partial, or because its previous sum cancelled to infinity -- or adds the entry to its accumulator.
        const bool start = (e == next_end) || acc_inf;
        if (start) {
            if (e == next_end) {
                // acc is a valid point here whatever the flag says (P + (-P) leaves a clean infinity behind)
                store_raw(partials + (size_t)(b + t) * RAW_WORDS, acc);
                b++;
                next_end = after_next;
                if (next_end <= e) {
                    // the next bucket is empty.  Not walked one by one: a bucket-range share has tens of thousands of empty buckets behind its
                    // last entry, skewed digits leave long runs between a handful of full ones, and a dependent load per empty bucket
                    // is ~90 ns (round 3 measurement: share 0 of 8 spent 5.4 ms here against 0.15 ms of additions).  Past the end of the
                    // list sits the dummy bucket; otherwise the bucket that holds entry e is found as at the start of the chunk.
                    if (e >= M) {
                        b = total_buckets;
                    } else {
                        uint32_t l2 = b, h2 = total_buckets; // gstart[l2] <= e < gstart[h2]
                        while (h2 - l2 > 1) {
                            const uint32_t mid = (l2 + h2) >> 1;
                            if (gstart[mid] <= e) l2 = mid; else h2 = mid;
                        }
                        b = l2;
                    }
                    next_end = gstart[b + 1];
                }
                after_next = gstart[min(b + 2, total_buckets + 1)];
            }
            acc.x = px;
            acc.y = py;
            acc.zz = fe_one<Fq>();
            acc.zzz = fe_one<Fq>();
            acc_inf = false;
        }
        asm volatile("" ::: "memory");
        if (!start) madd_ip(acc, acc_inf, px, py);
#ifdef BBGPU_ACC_JUNK // issue-model experiment (DESIGN_HISTORY 5): extra cheap VALU instructions per trip, results unused
        {
            uint32_t j0 = e, j1 = vn;
#pragma unroll
            for (int q = 0; q < BBGPU_ACC_JUNK / 2; q++) asm volatile("v_and_b32 %0, 0x1fffffff, %1\n\tv_add_u32 %1, %0, %1" : "+v"(j0), "+v"(j1));
        }
#endif
#ifdef BBGPU_ACC_JUNKMAD // the same with dependent v_mad_u64_u32
        {
            unsigned long long ja = e;
#pragma unroll
            for (int q = 0; q < BBGPU_ACC_JUNKMAD; q++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(ja) : "v"(vn) : "vcc");
        }
#endif
        load_affine_m261_signed(px, py, w, (vn >> 31) != 0);
        ld16(srs + (size_t)(vnn & 0x7fffffffu) * 16, w);
        vn = vnn;
        e++;
    } while (--trips != 0);
    store_raw(partials + (size_t)(b + t) * RAW_WORDS, acc);
}

// K4m: bucket b = sum of its partials, slots b + floor(s/ch) .. b + floor((e-1)/ch); written canonical for K5.
// 2^logG lanes share a bucket: lane j adds partials j, j + G, ... and the group is combined by a shuffle tree, so the dependent
// chain is ~ceil(count / G) + logG additions instead of `count` (one XYZZ addition is ~8 us of issue time for a lone wave:
// a 2^16-point MSM against window tables cuts every bucket into ~90 partials).  Buckets cut into more than MERGE_LIGHT
// partials (skewed digit distributions) are queued and summed by a whole workgroup each (K4h).
__device__ __forceinline__ Xyzz shfl_down_xyzz(const Xyzz& p, uint32_t off)
{
    Xyzz r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        r.x.d[i] = __shfl_down(p.x.d[i], off);
        r.y.d[i] = __shfl_down(p.y.d[i], off);
        r.zz.d[i] = __shfl_down(p.zz.d[i], off);
        r.zzz.d[i] = __shfl_down(p.zzz.d[i], off);
    }
    return r;
}
// Workgroup tree sum over up to 256 XYZZ points held one per lane (raw lazy limbs staged through LDS, structure of arrays, only
// the upper half of each level is stored: 128 x 144 B = 18 KiB): the dependent chain is log2(T) additions with no launch gaps.
// Result in lane 0.  Measured alternatives: a ds_bpermute shuffle tree (576 B of LDS) is 40-60 % slower per level (36 permutes
// per point and level); staging all 256 lanes needed 36 KiB, which beside three resident accumulation workgroups (3 x 41 KiB
// of the CU's 160 KiB) only fits when the free LDS happens to be contiguous.
// The tail kernels are held to 128 VGPRs (amdgpu_waves_per_eu(4, 4), ~32 registers spilled): at the 143 they would otherwise
// take, their waves do not fit beside the three 128-VGPR accumulation waves per SIMD of the next MSM and the whole tail queued
// behind it (rocprof timeline: merge 0.5 ms and heavy-merge 0.58 ms in the two-deep pipeline against 0.05 ms alone).
// TAIL_OCC: the register cap of the tail kernels (see above).  -DBBGPU_TAIL_WAVES=k sets another occupancy target for A/B builds (2: up to 256 VGPRs, no spills).
#ifndef BBGPU_TAIL_WAVES
#define BBGPU_TAIL_WAVES 4
#endif
#define TAIL_OCC __attribute__((amdgpu_waves_per_eu(BBGPU_TAIL_WAVES, BBGPU_TAIL_WAVES)))
constexpr int FOLD_T = 256;
constexpr int FOLD_LDS_WORDS = (FOLD_T / 2) * RAW_WORDS;
__device__ __forceinline__ void wg_tree_sum(Xyzz& acc, uint32_t* sh, uint32_t T, uint32_t t)
{
    constexpr uint32_t STRIDE = FOLD_T / 2;
    for (uint32_t half = T >> 1; half >= 1; half >>= 1) {
        if (t >= half && t < 2 * half) {
#pragma unroll
            for (int i = 0; i < NL; i++) {
                sh[(0 * NL + i) * STRIDE + t - half] = acc.x.d[i];
                sh[(1 * NL + i) * STRIDE + t - half] = acc.y.d[i];
                sh[(2 * NL + i) * STRIDE + t - half] = acc.zz.d[i];
                sh[(3 * NL + i) * STRIDE + t - half] = acc.zzz.d[i];
            }
        }
        __syncthreads();
        if (t < half) {
            Xyzz q, r;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                q.x.d[i] = sh[(0 * NL + i) * STRIDE + t];
                q.y.d[i] = sh[(1 * NL + i) * STRIDE + t];
                q.zz.d[i] = sh[(2 * NL + i) * STRIDE + t];
                q.zzz.d[i] = sh[(3 * NL + i) * STRIDE + t];
            }
            add(r, acc, q);
            acc = r;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(MSM_THREADS) TAIL_OCC msm_merge_kernel(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ partials,
                                                              uint32_t* __restrict__ buckets, uint32_t* __restrict__ heavy, uint32_t bucket_begin, uint32_t total_buckets,
                                                              uint32_t ch, uint32_t MERGE_LIGHT, uint32_t logG)
{
    // buckets [bucket_begin, total_buckets): a bucket-range share merges (and later folds) its own buckets only
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t G = 1u << logG, b = bucket_begin + (t >> logG), j = t & (G - 1);
    if (b >= total_buckets) return; // whole groups leave together (groups are aligned inside a wave)
    const uint32_t s = gstart[b], e = gstart[b + 1];
    Xyzz acc;
    set_infinity(acc);
    if (e > s) {
        const uint32_t t0 = s / ch, t1 = (e - 1) / ch;
        if (t1 - t0 >= MERGE_LIGHT) { // heavy[0] = count, heavy[HEAVY_IDS ..] = bucket ids
            if (j == 0) heavy[HEAVY_IDS + atomicAdd(&heavy[0], 1u)] = b;
            return;
        }
        for (uint32_t k = t0 + j; k <= t1; k += G) {
            Xyzz q, r;
            load_raw(q, partials + (size_t)(b + k) * RAW_WORDS);
            add(r, acc, q);
            acc = r;
        }
    }
    for (uint32_t off = G >> 1; off >= 1; off >>= 1) {
        const Xyzz o = shfl_down_xyzz(acc, off);
        if (j < off) {
            Xyzz r;
            add(r, acc, o);
            acc = r;
        }
    }
    if (j == 0) {
        uint32_t o[32];
        store_xyzz(o, acc);
        st32(buckets + (size_t)b * 32, o);
    }
}
// K4h: the queued (heavy) buckets.  Many of them: one workgroup per bucket (strided in-lane sums, then the workgroup tree).  FEW of them
// -- skewed digits: every scalar equal puts 2^20 entries, ~13 k partials, into each of 15 buckets -- : the gridDim.x workgroups are dealt
// K = gridDim.x / count to a bucket, each sums a slice of its partials into a raw partial of its own (hpart), and the LAST of a bucket's
// workgroups to arrive (agent-scope fences around an arrival counter) adds the K slices up (round 3: 0.45 -> ~0.1 ms for that case).
__global__ void __launch_bounds__(MSM_THREADS) TAIL_OCC msm_merge_heavy_kernel(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ partials,
                                                                    uint32_t* __restrict__ buckets, uint32_t* __restrict__ heavy, uint32_t ch, uint32_t* __restrict__ hpart)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    __shared__ uint32_t sh[FOLD_LDS_WORDS];
    __shared__ uint32_t last_flag;
    const uint32_t count = heavy[0];
    const uint32_t K = count ? gridDim.x / count : 0;
    if (K < 2) {
        for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
            const uint32_t b = heavy[HEAVY_IDS + item];
            const uint32_t s = gstart[b], e = gstart[b + 1];
            const uint32_t t0 = s / ch, t1 = (e - 1) / ch;
            Xyzz acc;
            set_infinity(acc);
            for (uint32_t t = t0 + threadIdx.x; t <= t1; t += MSM_THREADS) {
                Xyzz q, r;
                load_raw(q, partials + (size_t)(b + t) * RAW_WORDS);
                add(r, acc, q);
                acc = r;
            }
            __syncthreads(); // the previous item's wave 0 is done reading sh
            wg_tree_sum(acc, sh, MSM_THREADS, threadIdx.x);
            if (threadIdx.x == 0) {
                uint32_t o[32];
                store_xyzz(o, acc);
                st32(buckets + (size_t)b * 32, o);
            }
        }
        return;
    }
    const uint32_t item = blockIdx.x / K, sub = blockIdx.x - item * K;
    if (item >= count) return; // the gridDim.x - K * count workgroups left over
    const uint32_t b = heavy[HEAVY_IDS + item];
    const uint32_t s = gstart[b], e = gstart[b + 1];
    const uint32_t t0 = s / ch, t1 = (e - 1) / ch, total = t1 - t0 + 1;
    const uint32_t lo = t0 + (uint32_t)(((uint64_t)total * sub) / K), hi = t0 + (uint32_t)(((uint64_t)total * (sub + 1)) / K); // slice [lo, hi)
    Xyzz acc;
    set_infinity(acc);
    for (uint32_t t = lo + threadIdx.x; t < hi; t += MSM_THREADS) {
        Xyzz q, r;
        load_raw(q, partials + (size_t)(b + t) * RAW_WORDS);
        add(r, acc, q);
        acc = r;
    }
    wg_tree_sum(acc, sh, MSM_THREADS, threadIdx.x);
    if (threadIdx.x == 0) {
        store_raw(hpart + (size_t)blockIdx.x * RAW_WORDS, acc);
        __threadfence(); // the slice sum is visible device-wide before this workgroup is counted as arrived
        last_flag = (atomicAdd(&heavy[1 + item], 1u) == K - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;
    __threadfence(); // acquire: the other workgroups' slice sums
    set_infinity(acc);
    if (threadIdx.x < K) load_raw(acc, hpart + (size_t)(item * K + threadIdx.x) * RAW_WORDS);
    __syncthreads(); // wave 0 is done with sh from the first tree
    wg_tree_sum(acc, sh, MSM_THREADS, threadIdx.x);
    if (threadIdx.x == 0) {
        uint32_t o[32];
        store_xyzz(o, acc);
        st32(buckets + (size_t)b * 32, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K5: sum_b (b + 1) B_b over a bucket set without a serial running sum
// ---------------------------------------------------------------------------------------------------------------------
// m-th index with bit k set
__device__ __forceinline__ uint32_t insert_one_bit(uint32_t m, uint32_t k)
{
    return ((m >> k) << (k + 1)) | (1u << k) | (m & ((1u << k) - 1));
}

// Row sums R[hi] = sum_lo B[hi][lo] (blockIdx.x < H) and column sums C[lo] = sum_hi B[hi][lo] (blockIdx.x >= H) of the
// H x L bucket matrix of group blockIdx.y; blockDim.x = max(H, L).
__global__ void __launch_bounds__(FOLD_T) TAIL_OCC msm_rowcol_kernel(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ R, uint32_t* __restrict__ Cc,
                                                          uint32_t H, uint32_t L, uint32_t* __restrict__ zero_out, uint32_t zero_words)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    __shared__ uint32_t sh[FOLD_LDS_WORDS];
    const uint32_t g = blockIdx.y, t = threadIdx.x, nb = H * L;
    if (zero_out) { // small MSMs: the export slots (infinity = all zero) are cleared here instead of by a fill launch
        const uint32_t gid = (g * gridDim.x + blockIdx.x) * blockDim.x + t, all = gridDim.y * gridDim.x * blockDim.x;
        for (uint32_t i = gid; i < zero_words; i += all) zero_out[i] = 0;
    }
    const bool row = blockIdx.x < H;
    const uint32_t idx = row ? blockIdx.x : blockIdx.x - H, count = row ? L : H;
    Xyzz acc;
    set_infinity(acc);
    if (t < count) {
        const size_t b = row ? (size_t)idx * L + t : (size_t)t * L + idx;
        uint32_t w[32];
        ld32(buckets + ((size_t)g * nb + b) * 32, w);
        load_xyzz(acc, w);
    }
    wg_tree_sum(acc, sh, blockDim.x, t);
    if (t == 0) {
        uint32_t w[32];
        store_xyzz(w, acc);
        st32((row ? R + ((size_t)g * H + idx) * 32 : Cc + ((size_t)g * L + idx) * 32), w);
    }
}
// Job 0: Z = sum R; job 1 + k: TR_k = sum of the R_hi whose bit k is set; job 1 + hbits + k: TC_k likewise over C.  Each job is
// one workgroup; results go straight into the 64-slot export array in the reference's Montgomery form (slot 0 = Z,
// 1 + k = TR_k, 32 + k = TC_k; the array is zeroed = infinity beforehand).
__global__ void __launch_bounds__(FOLD_T) TAIL_OCC msm_final_kernel(const uint32_t* __restrict__ R, const uint32_t* __restrict__ Cc, uint32_t* __restrict__ out,
                                                         uint32_t hbits, uint32_t lbits)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    __shared__ uint32_t sh[FOLD_LDS_WORDS];
    const uint32_t g = blockIdx.y, t = threadIdx.x, job = blockIdx.x;
    const uint32_t H = 1u << hbits, L = 1u << lbits;
    const uint32_t* src;
    uint32_t count, slot, k = 0;
    bool sliced = true;
    if (job == 0) { src = R + (size_t)g * H * 32; count = H; slot = 0; sliced = false; }
    else if (job < 1 + hbits) { k = job - 1; src = R + (size_t)g * H * 32; count = H >> 1; slot = 1 + k; }
    else { k = job - 1 - hbits; src = Cc + (size_t)g * L * 32; count = L >> 1; slot = 32 + k; }
    Xyzz acc;
    set_infinity(acc);
    if (t < count) {
        uint32_t w[32];
        ld32(src + (size_t)(sliced ? insert_one_bit(t, k) : t) * 32, w);
        load_xyzz(acc, w);
    }
    wg_tree_sum(acc, sh, blockDim.x, t);
    if (t == 0) {
        uint32_t o[32];
        store_xyzz_m256(o, acc);
        st32(out + ((size_t)g * 64 + slot) * 32, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K5 with quad additions (g1_quad.hpp): the same two kernels, every point spread over the four lanes of a quad.  A workgroup of 256
// threads = 64 quads sums up to 256 points: each quad first adds its points e, e + 64, ... (<= 3 dependent additions), then a
// 4-level tree inside the wave (partners move by ds_bpermute: 9 words per lane and level, not 36), then the four wave sums through
// 576 bytes of LDS and two more levels: <= 9 dependent quad additions of ~1,250 instructions instead of 8 of ~3,700 plus a 36-word
// LDS round trip and two barriers per level.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int QFOLD_T = 256;
__device__ __forceinline__ FqN quad_load(const uint32_t* point32, uint32_t l) // coordinate l of a stored XYZZ point (4 x 8 words, canonical Montgomery-261)
{
    const uint4* q = reinterpret_cast<const uint4*>(point32 + 8 * l);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    return FqN(unpack<Fq>(w));
}
__device__ __forceinline__ FqN quad_zero()
{
    return FqN(fe_zero<Fq>());
}
// acc: this lane's coordinate of its quad's partial sum; returns the workgroup's sum in quad 0 of wave 0 (lanes 0..3).
// active (block-uniform): only quads [0, active) hold anything but the point at infinity.  Tree levels that could only add infinities are skipped (round 5:
// a small MSM without window tables sums bucket sets of 4 .. 32 buckets -- 2 .. 8 points per row -- and used to run all six levels, ~4 us each, on zeros).
template <int T = QFOLD_T> __device__ __forceinline__ FqN block_quad_sum(FqN acc, uint32_t* sh /* [T / 64][4][NL] */, uint32_t t, uint32_t active = T / 4)
{
    const uint32_t l = t & 3, lane = t & 63, wave = t >> 6, qd = lane >> 2;
    const uint32_t in_wave = active > 16 * wave ? min(16u, active - 16 * wave) : 0u; // quads of this wave that hold data (wave-uniform)
    for (uint32_t off = 8; off >= 1; off >>= 1) {
        if (off >= in_wave) continue; // every partner quad qd + off is empty
        FqN o;
#pragma unroll
        for (int i = 0; i < NL; i++) o.d[i] = __shfl_down(acc.d[i], off * 4);
        if (qd < off) acc = quad_add(acc, o, l);
    }
    if constexpr (T == 64) return acc; // one wave: the tree above was all of it
    constexpr uint32_t NW = T / 64;
    const uint32_t waves = min(NW, (active + 15) / 16); // waves that hold data (block-uniform)
    if (waves <= 1) return acc;                        // wave 0 has it all
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < NL; i++) sh[(wave * 4 + l) * NL + i] = acc.d[i];
    }
    __syncthreads();
    if (wave == 0) {
        acc = quad_zero();
        if (qd < NW) {
#pragma unroll
            for (int i = 0; i < NL; i++) acc.d[i] = sh[(qd * 4 + l) * NL + i];
        }
        for (uint32_t off = NW >> 1; off >= 1; off >>= 1) {
            if (off >= waves) continue;
            FqN o;
#pragma unroll
            for (int i = 0; i < NL; i++) o.d[i] = __shfl_down(acc.d[i], off * 4);
            if (qd < off) acc = quad_add(acc, o, l);
        }
    }
    return acc;
}
__device__ __forceinline__ void quad_store8(uint32_t* dst8, const uint32_t (&w)[8])
{
    uint4* q = reinterpret_cast<uint4*>(dst8);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// K4m with quad additions: 2^logQ quads share a bucket; quad j adds partials j, j + Q, ... (coordinate l of a raw partial = 9 lazy limbs at
// word 9 l), the quads of a bucket are combined by a shuffle tree and quad 0 writes the bucket canonical, 8 words per lane.  The chain is
// ~ceil(count / Q) + logQ quad additions of ~1,250 instructions: at 2^20 (4 partials per bucket, Q = 1) 3 quad additions instead of
// 3 full ones, at 2^16 (9.5 partials, Q = 4) 4 instead of 6.
__global__ void __launch_bounds__(MSM_THREADS) TAIL_OCC msm_merge_quad_kernel(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ partials,
                                                                   uint32_t* __restrict__ buckets, uint32_t* __restrict__ heavy, uint32_t bucket_begin, uint32_t total_buckets,
                                                                   uint32_t ch, uint32_t MERGE_LIGHT, uint32_t logQ)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, l = t & 3, quad = t >> 2;
    const uint32_t Q = 1u << logQ, b = bucket_begin + (quad >> logQ), j = quad & (Q - 1);
    if (b >= total_buckets) return; // whole bucket groups leave together (4 Q lanes, aligned inside a wave)
    const uint32_t s = gstart[b], e = gstart[b + 1];
    FqN acc = quad_zero();
    if (e > s) {
        const uint32_t t0 = s / ch, t1 = (e - 1) / ch;
        if (t1 - t0 >= MERGE_LIGHT) { // heavy[0] = count, heavy[HEAVY_IDS ..] = bucket ids
            if (j == 0 && l == 0) heavy[HEAVY_IDS + atomicAdd(&heavy[0], 1u)] = b;
            return;
        }
        // the next partial is loaded before the addition of the current one starts (a quad's partials are Q slots apart)
        uint32_t k = t0 + j;
        if (k <= t1) {
            const uint32_t* src = partials + (size_t)(b + k) * RAW_WORDS + NL * l;
#pragma unroll
            for (int i = 0; i < NL; i++) acc.d[i] = src[i];
            FqN q = quad_zero();
            k += Q;
            if (k <= t1) {
                src = partials + (size_t)(b + k) * RAW_WORDS + NL * l;
#pragma unroll
                for (int i = 0; i < NL; i++) q.d[i] = src[i];
            }
            while (k <= t1) { // quad-uniform trip count
                FqN qn = quad_zero();
                const uint32_t kn = k + Q;
                if (kn <= t1) {
                    src = partials + (size_t)(b + kn) * RAW_WORDS + NL * l;
#pragma unroll
                    for (int i = 0; i < NL; i++) qn.d[i] = src[i];
                }
                acc = quad_add(acc, q, l);
                q = qn;
                k = kn;
            }
        }
    }
    for (uint32_t off = Q >> 1; off >= 1; off >>= 1) {
        FqN o;
#pragma unroll
        for (int i = 0; i < NL; i++) o.d[i] = __shfl_down(acc.d[i], off * 4);
        if (j < off) acc = quad_add(acc, o, l);
    }
    if (j == 0) {
        uint32_t w[8];
        to_canonical(acc, w);
        quad_store8(buckets + (size_t)b * 32 + 8 * l, w);
    }
}

__global__ void __launch_bounds__(QFOLD_T) TAIL_OCC msm_rowcol_quad_kernel(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ R, uint32_t* __restrict__ Cc,
                                                                  uint32_t H, uint32_t L, uint32_t* __restrict__ zero_out, uint32_t zero_words, uint32_t r0, uint32_t rows)
{
    // rows [r0, r0 + rows) of the H x L bucket matrix (all of them, or a bucket-range share's): blockIdx.x < rows sums row r0 + blockIdx.x,
    // the L blocks after them sum the columns over those rows.  R of the other rows is not written: the caller zeroed it (infinity).
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    __shared__ uint32_t sh[(QFOLD_T / 64) * 4 * NL];
    const uint32_t g = blockIdx.y, t = threadIdx.x, nb = H * L, l = t & 3, quad = t >> 2;
    if (zero_out) { // small MSMs: the export slots (infinity = all zero) are cleared here instead of by a fill launch
        const uint32_t gid = (g * gridDim.x + blockIdx.x) * blockDim.x + t, all = gridDim.y * gridDim.x * blockDim.x;
        for (uint32_t i = gid; i < zero_words; i += all) zero_out[i] = 0;
    }
    const bool row = blockIdx.x < rows;
    const uint32_t idx = row ? r0 + blockIdx.x : blockIdx.x - rows, count = row ? L : rows;
    FqN acc = quad_zero();
    bool first = true;
    for (uint32_t e = quad; e < count; e += QFOLD_T / 4) { // quad-uniform trip count
        const size_t b = row ? (size_t)idx * L + e : (size_t)(r0 + e) * L + idx;
        const FqN v = quad_load(buckets + ((size_t)g * nb + b) * 32, l);
        acc = first ? v : quad_add(acc, v, l);
        first = false;
    }
    acc = block_quad_sum(acc, sh, t, min(count, (uint32_t)QFOLD_T / 4));
    if (t < 4) {
        uint32_t w[8];
        to_canonical(acc, w);
        quad_store8((row ? R + ((size_t)g * H + idx) * 32 : Cc + ((size_t)g * L + idx) * 32) + 8 * l, w);
    }
}
// K5 for large bucket sets, in two steps that leave no lane idle.  The one-launch form above spends half of its issue slots on tree levels with
// most quads masked off (2^16 buckets: 15.4 k wave-level quad additions for 131 k useful ones = 53 %; 20.6 M VALU instructions, the largest
// item of an MSM after the accumulation and the one a 1/N share of a multi-GPU MSM pays in full).  Step 1: ONE LANE per segment of ROWCOL_SEG
// buckets of a row (lanes [0, H L / SEG)) or of a column (the L H / SEG lanes after them) adds them with the plain addition (3,700
// lane-instructions against 4 x 1,350 for a quad): 3 dependent additions, every lane busy.  Step 2: one wave per row / column sums its
// L / SEG (H / SEG) segment sums as 16 quads (<= 3 sequential + 4 tree levels).  2^16 buckets: 5.7 M + 4.5 M instructions for the same chain length.
constexpr uint32_t ROWCOL_SEG = 4;
__global__ void __launch_bounds__(MSM_THREADS) TAIL_OCC msm_rowcol_seg_kernel(const uint32_t* __restrict__ buckets, uint32_t* __restrict__ segs, uint32_t H, uint32_t L,
                                                                                                                uint32_t* __restrict__ zero_out, uint32_t zero_words)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO); // tail kernels: short dependent chains, see msm_issue()
    const uint32_t g = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x, nb = H * L;
    if (zero_out) { // the export slots (infinity = all zero) are cleared here instead of by a fill launch
        const uint32_t gid = g * gridDim.x * blockDim.x + t, all = gridDim.y * gridDim.x * blockDim.x;
        for (uint32_t i = gid; i < zero_words; i += all) zero_out[i] = 0;
    }
    const uint32_t row_lanes = H * (L / ROWCOL_SEG), col_lanes = L * (H / ROWCOL_SEG);
    if (t >= row_lanes + col_lanes) return;
    // rows: lane t = (row, s) sums buckets row * L + s + j * (L / SEG); columns: lane u = (s, col) sums rows s + j * (H / SEG) of column col.
    // Either way neighbouring lanes read neighbouring buckets.  Segment sums of row r: segs[r * (L / SEG) + s]; of column c: behind the rows, [c * (H / SEG) + s].
    size_t b, step, out;
    if (t < row_lanes) {
        const uint32_t per = L / ROWCOL_SEG, r = t / per, sidx = t - r * per;
        b = (size_t)r * L + sidx;
        step = per;
        out = t;
    } else {
        const uint32_t u = t - row_lanes, per = H / ROWCOL_SEG, sidx = u / L, c = u - sidx * L;
        b = (size_t)sidx * L + c;
        step = (size_t)per * L;
        out = (size_t)row_lanes + (size_t)c * per + sidx;
    }
    const uint32_t* src = buckets + ((size_t)g * nb + b) * 32;
    Xyzz acc;
    {
        uint32_t w[32];
        ld32(src, w);
        load_xyzz(acc, w);
    }
    for (uint32_t j = 1; j < ROWCOL_SEG; j++) {
        uint32_t w[32];
        ld32(src + j * step * 32, w);
        Xyzz q, r;
        load_xyzz(q, w);
        add(r, acc, q);
        acc = r;
    }
    uint32_t o[32];
    store_xyzz(o, acc);
    st32(segs + ((size_t)g * (row_lanes + col_lanes) + out) * 32, o);
}
__global__ void __launch_bounds__(64) TAIL_OCC msm_segsum_quad_kernel(const uint32_t* __restrict__ segs, uint32_t* __restrict__ R, uint32_t* __restrict__ Cc, uint32_t H, uint32_t L)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO);
    const uint32_t g = blockIdx.y, t = threadIdx.x, l = t & 3, quad = t >> 2;
    const uint32_t row_lanes = H * (L / ROWCOL_SEG), col_lanes = L * (H / ROWCOL_SEG);
    const bool row = blockIdx.x < H;
    const uint32_t idx = row ? blockIdx.x : blockIdx.x - H, count = row ? L / ROWCOL_SEG : H / ROWCOL_SEG;
    const uint32_t* src = segs + ((size_t)g * (row_lanes + col_lanes) + (row ? (size_t)idx * count : (size_t)row_lanes + (size_t)idx * count)) * 32;
    FqN acc = quad_zero();
    bool first = true;
    for (uint32_t e = quad; e < count; e += 16) { // quad-uniform trip count
        const FqN v = quad_load(src + (size_t)e * 32, l);
        acc = first ? v : quad_add(acc, v, l);
        first = false;
    }
    acc = block_quad_sum<64>(acc, nullptr, t, min(count, 16u));
    if (t < 4) {
        uint32_t w[8];
        to_canonical(acc, w);
        quad_store8((row ? R + ((size_t)g * H + idx) * 32 : Cc + ((size_t)g * L + idx) * 32) + 8 * l, w);
    }
}
__global__ void __launch_bounds__(QFOLD_T) TAIL_OCC msm_final_quad_kernel(const uint32_t* __restrict__ R, const uint32_t* __restrict__ Cc, uint32_t* __restrict__ out,
                                                                 uint32_t hbits, uint32_t lbits)
{
    __builtin_amdgcn_s_setprio(BBGPU_TAIL_PRIO);
    __shared__ uint32_t sh[(QFOLD_T / 64) * 4 * NL];
    const uint32_t g = blockIdx.y, t = threadIdx.x, job = blockIdx.x, l = t & 3, quad = t >> 2;
    const uint32_t H = 1u << hbits, L = 1u << lbits;
    const uint32_t* src;
    uint32_t count, slot, k = 0;
    bool sliced = true;
    if (job == 0) { src = R + (size_t)g * H * 32; count = H; slot = 0; sliced = false; }
    else if (job < 1 + hbits) { k = job - 1; src = R + (size_t)g * H * 32; count = H >> 1; slot = 1 + k; }
    else { k = job - 1 - hbits; src = Cc + (size_t)g * L * 32; count = L >> 1; slot = 32 + k; }
    FqN acc = quad_zero();
    bool first = true;
    for (uint32_t e = quad; e < count; e += QFOLD_T / 4) {
        const FqN v = quad_load(src + (size_t)(sliced ? insert_one_bit(e, k) : e) * 32, l);
        acc = first ? v : quad_add(acc, v, l);
        first = false;
    }
    acc = block_quad_sum(acc, sh, t, min(count, (uint32_t)QFOLD_T / 4));
    if (t < 4) {
        uint32_t w[8];
        to_canonical(m261_to_m256<Fq>(acc), w); // the host finishes in the reference's Montgomery form (store_xyzz_m256, coordinate by coordinate)
        quad_store8(out + ((size_t)g * 64 + slot) * 32 + 8 * l, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------------------
#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));                                \
            return BBGPU_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)

int msm_choose_c(size_t n)
{
    // (round 5 swept fixed window sizes 5 .. 10 for the table-less small MSMs, 32 .. 1000 points, against this rule: nothing better, profiles/r05_small_sizes.txt)
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    int c = lg - 4;
    if (c < 4) c = 4;
    if (c > MSM_MAX_C) c = MSM_MAX_C;
    return c;
}
int msm_num_windows(int c)
{
    return SCALAR_BITS / c + 1;
}

struct MsmPlan {
    uint32_t n, c, W, nb, hbits, lbits, slices, slice_len, sort_lb, sort_bins;
};
// K4 residency: 3 workgroups of 256 lanes per CU (3 waves per SIMD; v_mad_u64_u32 issue saturates at 2).  The 4th slot is
// deliberately left free -- enforced by a dynamic-LDS reservation -- so that the short latency-bound kernels of the
// previous MSM's tail (merge, folds) can run beside the accumulation of the next one (two-slot pipeline).
constexpr uint32_t ACC_LDS_RESERVE = 41 * 1024; // 3 x 41 KiB fit in 160 KiB, 4 do not
static uint32_t acc_wg_per_cu()
{
    static int v = 0;
    if (!v) {
        v = 3;
        if (const char* e = getenv("BBGPU_ACC_WGS")) v = std::min(4, std::max(1, atoi(e))); // tuning knob (4 = no reservation)
    }
    return (uint32_t)v;
}
static uint32_t acc_lds_reserve()
{
    static int v = -1;
    if (v < 0) {
        v = acc_wg_per_cu() == 3 ? (int)ACC_LDS_RESERVE : (acc_wg_per_cu() == 2 ? 60 * 1024 : 0);
        if (const char* e = getenv("BBGPU_ACC_LDS")) v = std::min(64 * 1024, std::max(0, atoi(e))); // tuning knob
    }
    return (uint32_t)v;
}
static uint32_t acc_prio()
{
    static int v = -1;
    if (v < 0) {
        v = 0;
        if (const char* e = getenv("BBGPU_ACC_PRIO")) v = std::min(2, std::max(0, atoi(e))); // tuning knob
    }
    return (uint32_t)v;
}
static uint32_t acc_capacity_lanes()
{
    static uint32_t lanes = 0;
    if (!lanes) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        lanes = (uint32_t)cus * acc_wg_per_cu() * MSM_THREADS;
    }
    return lanes;
}
// chunk length of K4: one resident wave of workgroups covers the whole entry list (no tail wave), >= MIN_CHUNK entries per lane
// (small MSMs are latency-bound: a lane's chain of `ch` dependent mixed additions is the critical path, ~5 us each)
constexpr uint32_t MIN_CHUNK = 8;
// ... 4 for entry lists so short that even then a fraction of the chip is busy (round 5: 256 points without window tables are 16 K entries = eight workgroups walking
// eight dependent mixed additions each, 45 us; with chunks of four 23 us for one more level in the merge)
constexpr uint64_t TINY_ENTRIES = (uint64_t)1 << 16;
constexpr uint32_t TINY_MIN_CHUNK = 4;
static uint32_t min_chunk(uint64_t m) { return m <= TINY_ENTRIES ? TINY_MIN_CHUNK : MIN_CHUNK; }
static uint32_t chunk_len_m(uint64_t m);
static bool chunk_forced()
{
    static const bool f = getenv("BBGPU_CHUNK") != nullptr || getenv("BBGPU_ACC_WAVES") != nullptr;
    return f;
}
static uint32_t chunk_len(size_t n, uint32_t nw)
{
    return chunk_len_m((uint64_t)n * nw);
}
// m: expected number of entries of the sorted list
static uint32_t chunk_len_m(const uint64_t m)
{
    static int waves = 0; // BBGPU_ACC_WAVES: k > 1 cuts the list into k times as many (shorter) chunks -> k waves of workgroups (tuning experiments)
    if (!waves) {
        waves = 1;
        if (const char* e = getenv("BBGPU_ACC_WAVES")) waves = std::min(16, std::max(1, atoi(e)));
    }
    static const int forced = [] { const char* e = getenv("BBGPU_CHUNK"); return e ? std::max((int)MIN_CHUNK, atoi(e)) : 0; }(); // tuning knob (small MSMs)
    if (forced) return (uint32_t)std::max<uint64_t>(forced, (m + ((uint64_t)1 << 24) - 1) >> 24);
    if (waves > 1) {
        const uint64_t cap = (uint64_t)acc_capacity_lanes() * (uint64_t)waves;
        return std::max<uint32_t>(MIN_CHUNK, (uint32_t)((m + cap - 1) / cap));
    }
    // k = 1 .. 3 workgroups per CU, every lane `ch` entries: the kernel lasts ~ch * step(k), step(k) = time of one mixed addition of a wave
    // with k waves on its SIMD: ~6.4 us alone (dependent multiplications), k * 4.46 us once two waves saturate the multiplier
    // (14.7e9 mixed additions per second chip-wide).  A grid that is NOT a whole number of workgroups per CU runs at the pace of the
    // fullest CU: 2^16 points x 17 windows at ch = 8 is 543 workgroups = 2.1 per CU, paced by the CUs holding 3 (107 us);
    // ch = 9 gives 484 = at most 2 per CU (80 us).
    static const double step[3] = { 6.4, 9.2, 13.4 }; // two waves reach ~97 % of the multiplier rate, three all of it
    const uint64_t per_k = (uint64_t)acc_capacity_lanes() / acc_wg_per_cu(); // lanes of one workgroup per CU
    uint32_t best = 0;
    double best_cost = 0.0;
    for (uint32_t k = 1; k <= acc_wg_per_cu() && k <= 3; k++) {
        const uint32_t ch = std::max<uint32_t>(min_chunk(m), (uint32_t)((m + per_k * k - 1) / (per_k * k)));
        const double cost = ch * step[k - 1];
        if (!best || cost < best_cost) { best = ch; best_cost = cost; }
    }
    return best;
}
static size_t arena_points(const MsmPlan& P, uint32_t nw)
{
    const size_t H = (size_t)1 << P.hbits, L = (size_t)1 << P.lbits;
    return (size_t)nw * (H + L) + 64; // row sums + column sums per bucket set
}
// two-step row / column sums (msm_rowcol_seg_kernel): bucket sets of 2^15 and more
static bool rowcol_two_step(const MsmPlan& P)
{
    return P.hbits + P.lbits >= 15;
}
static size_t seg_points(const MsmPlan& P, uint32_t nw)
{
    const size_t H = (size_t)1 << P.hbits, L = (size_t)1 << P.lbits;
    return rowcol_two_step(P) ? (size_t)nw * 2 * (H * L / ROWCOL_SEG) : 0;
}
static MsmPlan make_plan(size_t n, int c)
{
    MsmPlan P;
    P.n = (uint32_t)n;
    P.c = (uint32_t)c;
    P.W = (uint32_t)msm_num_windows(c);
    P.nb = 1u << (c - 1);
    P.lbits = (c - 1) / 2;
    P.hbits = (c - 1) - P.lbits;
    P.slices = std::max<uint32_t>(1, 512 / P.W);
    P.sort_lb = std::min<uint32_t>(SORT_MAX_LB, (uint32_t)c - 1);
    P.sort_bins = P.nb >> P.sort_lb; // <= 256 for c <= 16
    if ((uint64_t)P.slices * 4096 > n) P.slices = std::max<uint32_t>(1, (uint32_t)(n / 4096));
    P.slice_len = (uint32_t)((n + P.slices - 1) / P.slices);
    return P;
}

// The ONE description of the workspace: byte offsets of every array for `nw` (job, window) pairs.  Both the size request
// (MsmWorkspace::bytes_needed) and the carve in msm_issue_batch read it, and the issue path checks `end` against the
// allocation before anything is launched.  (Round 1's only GPU memory fault -- gpurun_out/quick1.txt, first 2^20 bring-up --
// came from exactly such a pair of hand-kept formulas: the fold arena was sized by a guess, (2 nw nb + 4096) points, while
// the then per-level fold / slice chain bumped rows + columns + slices per window past it into the next page.)
struct MsmCarve {
    size_t digits, signs, histA, histB, binstart, bintot, tmp_entries, gstart, totals, heavy, sorted, partials, buckets, arena, segs, texp, end;
    size_t chunks_cap, histB_bytes;
};
// nw: (job, window) pairs = what the entry lists scale with; ng: BUCKET SETS = what the bucket-side arrays scale with -- nw without window tables (one set per
// window), the number of jobs with them (every window feeds the job's one shared set).  Round 5: the bucket-side arrays were sized by nw in both modes, 15 times
// what a 2^20-point MSM against window tables touches (0.85 GB per slot instead of 0.5).
static MsmCarve carve(const MsmPlan& P, size_t n, size_t nw, size_t ng)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    MsmCarve L;
    size_t p = 0;
    const size_t wmax = std::max<size_t>(P.W, nw); // a batch of j jobs passes nw = j * W
    L.digits = p;      p += al(wmax * n * 2);
    L.signs = p;       p += al(wmax * ((n + 63) / 64) * 8);      // sign bits of 17-bit windows
    L.histA = p;       p += al(nw * P.slices * 1024 * 4);        // pass-A histogram / cursors (<= 1024 bins)
    // pass-B piece counts: read only when heavy bins are cut into pieces (table mode, n * windows >= 2^21 digits, <= MSM_MAX_JOBS groups): nothing for smaller MSMs
    L.histB_bytes = (uint64_t)n * nw >= ((uint64_t)1 << 21) ? (size_t)MSM_MAX_JOBS * 1024 * SORTB_MAX_PARTS * 128 * 4 : 0;
    L.histB = p;       p += al(L.histB_bytes);
    L.binstart = p;    p += al(nw * 1024 * 4 + 256);
    L.bintot = p;      p += al(nw * 1024 * 4 + 256);
    L.tmp_entries = p; p += al(nw * n * 4);                      // pass-A output
    L.gstart = p;      p += al((ng * P.nb + 2) * 4);             // + M at [total_buckets] + a sentinel behind it
    L.totals = p;      p += al(nw * 8 + 512);                    // totals, bases (nw + 1)
    L.heavy = p;       p += al((ng * P.nb + HEAVY_IDS) * 4 + 256 + (size_t)HEAVY_WGS * RAW_WORDS * 4); // heavy-bucket queue: count, arrival counters, ids, slice sums
    L.sorted = p;      p += al(nw * n * 4);
    L.chunks_cap = (n * nw + min_chunk(n * nw) - 1) / min_chunk(n * nw) + 1; // upper bound for any chunk length >= min_chunk (of the WHOLE list: a share's list is shorter, its chunks no shorter than this)
    L.partials = p;    p += al((ng * P.nb + L.chunks_cap) * RAW_WORDS * 4);
    L.buckets = p;     p += al(ng * P.nb * 128);
    L.arena = p;       p += al(arena_points(P, (uint32_t)ng) * 128); // row sums + column sums
    L.segs = p;        p += al(seg_points(P, (uint32_t)ng) * 128);   // segment sums of the two-step row / column sums (large bucket sets)
    L.texp = p;        p += al(ng * 64 * 128);                   // exported T points
    L.end = p;
    return L;
}
size_t MsmWorkspace::bytes_needed(size_t n, int c, int nw, int ng)
{
    return carve(make_plan(n, c), n, (size_t)nw, (size_t)(ng > 0 ? ng : nw)).end;
}

int MsmWorkspace::ensure(size_t bytes)
{
    if (bytes <= cap) return BBGPU_OK;
    if (base) (void)dev_free(base);
    base = nullptr;
    cap = 0;
    HIPCHK(dev_malloc((void**)&base, bytes));
    cap = bytes;
    return BBGPU_OK;
}
void MsmWorkspace::release()
{
    if (base) (void)dev_free(base);
    if (h_out) (void)hipHostFree(h_out);
    base = nullptr;
    h_out = nullptr;
    cap = 0;
}

// Runs windows [wb, we) of the MSM of d_scalars[0..n) against resident points srs[0..n); returns the partial sum
// sum_{w in [wb,we)} 2^(c w) S_w as host XYZZ (Montgomery 2^256).
// Enqueues windows [wb, we) of the MSM of d_scalars[0..n) against resident points srs[0..n) on `st` (all kernels and the
// final 16 KiB device-to-host copy of the per-window leftover points); returns without waiting.  msm_finish() waits for
// the slot's event and runs the host tail.  Two slots let the tail of one MSM overlap the head of the next.
// Table mode (d_tab != nullptr): d_tab[w * tab_stride + i] = 2^(tab_c * w) * P_i for the points of this call, so every window
// feeds ONE shared bucket set (groups = 1): the bucket reduction and the host finish shrink 16-fold and no positional
// doublings are needed.
int msm_issue(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n, int wb,
              int we, hipStream_t st, int want_timing)
{
    return msm_issue_batch(S, d_srs, d_tab, tab_stride, tab_c, &d_scalars, 1, n, wb, we, st, want_timing, 0, (uint32_t)n);
}
// Share [row_begin, row_end) of the W * n (window, point) pairs, rows counted window-major: row = w * n + i.  With window tables every pair
// is just one table row feeding the one shared bucket set, so ANY split of the rows is a valid split of the MSM; splitting rows instead
// of whole windows keeps N ranks balanced when N does not divide W (15 windows over 8 ranks: 1.875 windows each instead of 2).
int msm_issue_rows(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n,
                   uint64_t row_begin, uint64_t row_end, hipStream_t st, int want_timing)
{
    if (!d_tab || n == 0 || row_end <= row_begin || row_end > (uint64_t)msm_num_windows(tab_c) * n) return BBGPU_ERR_ARG;
    const int wb = (int)(row_begin / n), we = (int)((row_end + n - 1) / n);
    const uint32_t i0 = (uint32_t)(row_begin - (uint64_t)wb * n), i1 = (uint32_t)(row_end - (uint64_t)(we - 1) * n);
    return msm_issue_batch(S, d_srs, d_tab, tab_stride, tab_c, &d_scalars, 1, n, wb, we, st, want_timing, i0, i1);
}

// One of `share_count` BUCKET-RANGE shares of an MSM against window tables: every share reads all digits of all windows but keeps only
// those whose bucket lies in its rows [H s / N, H (s + 1) / N) of the 2^hbits x 2^lbits bucket matrix, i.e. 1 / N of the mixed
// additions AND 1 / N of the buckets to merge and fold -- the part of an MSM that a row-range share (above) repeats in full on every
// rank.  The share's result is sum_{b in range} (b + 1) B_b (the bit-slice weights use the global bucket index), so the shares of
// one MSM add up to it.  Uniform digits give equal shares; skewed ones (all scalars equal) put whole windows into one share.
int msm_issue_buckets(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* d_scalars, size_t n,
                      uint32_t share, uint32_t share_count, hipStream_t st, int want_timing)
{
    if (!d_tab || n == 0 || share_count == 0 || share >= share_count) return BBGPU_ERR_ARG;
    const MsmPlan P = make_plan(n, tab_c);
    const uint32_t H = 1u << P.hbits;
    if (share_count > H) return BBGPU_ERR_ARG;
    const uint32_t r0 = (uint32_t)((uint64_t)H * share / share_count), r1 = (uint32_t)((uint64_t)H * (share + 1) / share_count);
    return msm_issue_batch(S, d_srs, d_tab, tab_stride, tab_c, &d_scalars, 1, n, 0, (int)P.W, st, want_timing, 0, 0xffffffffu, r0, r1);
}

// `jobs` MSMs of n scalars each over the SAME points as one pass through the pipeline (the prover's 3 / 1 / 3 / 2
// commitments per round, prover.cpp:65-122,650-658): each job is a bucket set ("group") of the shared sort / accumulate / merge /
// reduction kernels, so the batch costs one chain of launches and one chain of dependent group additions instead of `jobs`.
// Table mode and the full window range only.  msm_finish_batch returns one point per job.
// "accumulation ended" events of the last ACC_RING timed MSMs, in issue order.  With two MSMs in flight the next accumulation is
// enqueued while the previous one still runs (its sort is done early), so the event pair around the kernel also measures the time it sat
// in the queue; the kernel cannot execute before the previous accumulation has drained (one resident wave of workgroups fills the chip),
// so its execution time is at most the spacing of consecutive "ended" events.
constexpr int ACC_RING = 8;
static hipEvent_t g_acc_end[ACC_RING];
static uint64_t g_acc_seq = 0; // number of timed accumulations issued so far
static int acc_ring_record(MsmSlot& S, hipStream_t st)
{
    if (g_acc_seq == 0)
        for (int i = 0; i < ACC_RING; i++) HIPCHK(hipEventCreate(&g_acc_end[i]));
    S.acc_seq = ++g_acc_seq;
    HIPCHK(hipEventRecord(g_acc_end[S.acc_seq % ACC_RING], st));
    return BBGPU_OK;
}
int msm_issue_batch(MsmSlot& S, const uint32_t* d_srs, const uint32_t* d_tab, size_t tab_stride, int tab_c, const uint64_t* const* d_scalars_v, int jobs,
                    size_t n, int wb, int we, hipStream_t st, int want_timing, uint32_t row_i0, uint32_t row_i1, uint32_t brow0, uint32_t brow1)
{
    MsmWorkspace& ws = S.ws;
    // a further piece of the MSM already issued on this slot (same jobs, same stream: stream order hands the workspace on), or a new MSM
    const bool append = S.append && S.pending && S.npieces > 0;
    S.append = false;
    if (append && (S.jobs != (uint32_t)jobs || S.npieces >= MSM_MAX_PIECES)) {
        set_error("internal: MSM piece %d does not continue the MSM on this slot", S.npieces);
        return BBGPU_ERR_STATE;
    }
    if (!append) {
        S.jobs = (uint32_t)jobs;
        S.n = n;
        S.pending = false;
        S.timed = false;
        S.npieces = 0;
    } else {
        S.n += n;
    }
    MsmPiece& PC = S.piece[S.npieces];
    PC = MsmPiece{};
    PC.hout_group = S.npieces ? S.piece[S.npieces - 1].hout_group + S.piece[S.npieces - 1].nw : 0u;
    if (n == 0) { PC.trivial = true; S.npieces++; S.pending = true; return BBGPU_OK; }
    if (n > ((size_t)1 << 24)) { // sort entries carry a 24-bit point index
        set_error("MSM of %zu points: at most 2^24 points per call", n);
        return BBGPU_ERR_SIZE;
    }
    const bool hint = S.throughput; // one-shot: the synchronous entry points use slots 0 / 1 without going through pick_slot()
    S.throughput = false;
    const bool table = d_tab != nullptr;
    const int c = table ? tab_c : msm_choose_c(n);
    const MsmPlan P = make_plan(n, c);
    if (wb < 0 || we > (int)P.W || wb >= we) return BBGPU_ERR_ARG;
    if (row_i1 > n) row_i1 = (uint32_t)n;
    // a share that starts / ends inside a window (msm_issue_rows): only with one shared bucket set, i.e. against window tables
    if ((row_i0 != 0 || row_i1 != n) && (!table || jobs != 1 || row_i0 >= n || (we - wb == 1 && row_i0 >= row_i1))) return BBGPU_ERR_ARG;
    if (jobs < 1 || jobs > MSM_MAX_JOBS || (jobs > 1 && (!table || wb != 0 || we != (int)P.W))) {
        set_error("batched MSM: 1..%d jobs, window tables and the full window range required", MSM_MAX_JOBS);
        return BBGPU_ERR_ARG;
    }
    // a bucket-range share (msm_issue_buckets): rows [brow0, brow1) of the 2^hbits x 2^lbits bucket matrix over ALL windows and points --
    // like a row-range share only with one shared bucket set (window tables), one job, and quad tail kernels
    const uint32_t BH = 1u << P.hbits, BL = 1u << P.lbits;
    if (brow1 > BH) brow1 = BH;
    const bool bshare = brow0 != 0 || brow1 != BH;
    if (bshare && (!table || jobs != 1 || brow0 >= brow1 || wb != 0 || we != (int)P.W || row_i0 != 0 || row_i1 != n)) return BBGPU_ERR_ARG;
    const uint32_t blo = brow0 * BL, bcnt = (brow1 - brow0) * BL;
    const uint32_t nw1 = (uint32_t)(we - wb);     // windows processed per job
    const uint32_t nw = nw1 * (uint32_t)jobs;     // (job, window) pairs: what the entry count and the workspace scale with
    const uint32_t G = table ? (uint32_t)jobs : nw1; // bucket sets ("groups")
    const uint32_t wpg = table ? nw1 : 1u;        // windows per group
    // one shared bucket set holds nw times the entries: finer bins (<= 1024) keep pass B's per-workgroup share small
    uint32_t sort_lb = P.sort_lb, sort_bins = P.sort_bins;
    if (table) {
        uint32_t want_bins = 512; // measured: 512 bins 0.196 ms, 1024 bins 0.229 ms, 256 bins 0.215 ms (2^20, 256 slices)
        if (const char* e = getenv("BBGPU_SORT_BINS")) want_bins = std::min(1024, std::max(64, atoi(e))); // tuning knob
        while (sort_bins < want_bins && sort_lb > 3) { sort_lb--; sort_bins <<= 1; }
    }
    uint32_t slices = table ? std::max<uint32_t>(1, P.slices * nw1 / 2) : P.slices;
    // a share of a few windows (a rank of an N-way split) would run pass A on a few dozen workgroups: rocprofv3 timeline of a 1/8 row share,
    // 51 slices: histogram 12-46 us, staged scatter 36-79 us for 2 M entries (the whole 2^20 MSM: 52 us for 15.7 M).  At least ~200 slices
    // where the points allow >= 2048 per slice and the histogram matrix of the workspace (nw * P.slices * 1024 words) holds them.
    if (table && jobs == 1) {
        const uint32_t by_cap = (uint32_t)(((uint64_t)nw * P.slices * 1024) / ((uint64_t)G * sort_bins));
        const uint32_t want = std::min<uint32_t>(std::min<uint32_t>(208, (uint32_t)(n / 2048)), by_cap);
        slices = std::max(slices, want);
    }
    if (const char* e = getenv("BBGPU_SLICES")) slices = std::min<uint32_t>(std::max(1, atoi(e)), P.slices * nw1); // tuning knob
    const uint32_t slice_len = (uint32_t)((n + slices - 1) / slices);
    const uint32_t idx_stride = table ? (uint32_t)tab_stride : 0u;
    const uint32_t* points = table ? d_tab : d_srs;
    PC.c = P.c; PC.nw = G; PC.wb = table ? 0u : (uint32_t)wb; PC.hbits = P.hbits; PC.lbits = P.lbits;
    if (PC.hout_group + G > MSM_HOUT_GROUPS) {
        set_error("MSM in %d pieces of %u bucket sets: the result array holds %u", S.npieces + 1, G, MSM_HOUT_GROUPS);
        return BBGPU_ERR_SIZE;
    }
    // (capi.hip issue_ticket() sizes the workspace for the LARGEST piece of a multi-piece MSM before piece 0 is issued -- a slice that starts mid-segment has
    // its largest piece in the middle -- so that this never frees a workspace under a piece still queued on the stream)
    int rc = ws.ensure(MsmWorkspace::bytes_needed(n, c, (int)nw, (int)G));
    if (rc) return rc;
    if (!ws.h_out) HIPCHK(hipHostMalloc((void**)&ws.h_out, (size_t)MSM_HOUT_GROUPS * 64 * 128));
    if (!S.done) HIPCHK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));

    const MsmCarve LY = carve(P, n, nw, G);
    if (LY.end > ws.cap) { // cannot happen while bytes_needed and this carve share one table; checked before anything is launched
        set_error("internal: MSM workspace of %zu bytes, layout needs %zu (n=%zu c=%d nw=%u)", ws.cap, LY.end, n, c, nw);
        return BBGPU_ERR_STATE;
    }
    uint8_t* const p = ws.base;
    void* digits = (void*)(p + LY.digits);
    unsigned long long* signs = (unsigned long long*)(p + LY.signs);
    uint32_t* histA = (uint32_t*)(p + LY.histA);
    uint32_t* histB = (uint32_t*)(p + LY.histB);
    uint32_t* binstart = (uint32_t*)(p + LY.binstart);
    uint32_t* bintot = (uint32_t*)(p + LY.bintot);
    uint32_t* tmp_entries = (uint32_t*)(p + LY.tmp_entries);
    uint32_t* gstart = (uint32_t*)(p + LY.gstart);
    uint32_t* totals = (uint32_t*)(p + LY.totals);
    uint32_t* bases = totals + nw;
    uint32_t* heavy = (uint32_t*)(p + LY.heavy);
    uint32_t* sorted = (uint32_t*)(p + LY.sorted);
    uint32_t* partials = (uint32_t*)(p + LY.partials);
    uint32_t* buckets = (uint32_t*)(p + LY.buckets);
    uint32_t* scratch = (uint32_t*)(p + LY.arena);
    uint32_t* texp = (uint32_t*)(p + LY.texp);
    // launch-shape checks against the carve (the kernels index these arrays from these quantities)
    if ((uint64_t)G * slices * sort_bins > (uint64_t)nw * P.slices * 1024 || sort_bins > 1024 ||
        (size_t)G * ((size_t)(1u << P.hbits) + (size_t)(1u << P.lbits)) + 64 > arena_points(P, G)) {
        set_error("internal: MSM launch shape exceeds the workspace layout");
        return BBGPU_ERR_STATE;
    }

    hipEvent_t* ev = S.ev;
    // timing level 1: an event after every stage (eight markers per MSM: each one is a dependent packet in the stream, ~10 us of latency
    // on the front / tail chains -- measured 0.085 ms per pipelined 2^20 step); level 2: only the pair around the accumulation, which is
    // what bench.py's timed region records
    const bool tm = want_timing == 1, tm_acc = want_timing != 0;
    S.timed_light = want_timing == 2;
    if (tm_acc) {
        if (!S.ev[0])
            for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&S.ev[i]));
        if (tm) HIPCHK(hipEventRecord(ev[0], st));
        S.timed = true;
    }

    if (bshare) {
        static const bool quad_ok = [] { const char* e = getenv("BBGPU_QUAD_TAIL"); return !e || atoi(e) != 0; }();
        if (!quad_ok) {
            set_error("bucket-range shares need the quad tail kernels (BBGPU_QUAD_TAIL=0 is set)");
            return BBGPU_ERR_STATE;
        }
        // row sums outside the share stay infinity (all zero): the bit-slice kernel reads every row.  First in the stream, far off the tail's critical path
        HIPCHK(hipMemsetAsync(scratch, 0, (size_t)G * BH * 128, st));
    }
    // K0
    ScalarSets sets{};
    for (int j = 0; j < jobs; j++) sets.p[j] = (const uint32_t*)d_scalars_v[j];
    const bool wide = c > 16; // 17-bit windows: signed digits up to +-2^16
    if (wide) msm_digits_kernel<uint16_t><<<dim3((P.n + MSM_THREADS - 1) / MSM_THREADS, jobs), MSM_THREADS, 0, st>>>(sets, (uint16_t*)digits, signs, P.n, make_layout(c, table), P.W, (uint32_t)wb, (uint32_t)we);
    else msm_digits_kernel<int16_t><<<dim3((P.n + MSM_THREADS - 1) / MSM_THREADS, jobs), MSM_THREADS, 0, st>>>(sets, (int16_t*)digits, signs, P.n, make_layout(c, table), P.W, (uint32_t)wb, (uint32_t)we);
    if (tm) HIPCHK(hipEventRecord(ev[1], st));
    // K1-K3
    if (wide) sortA_hist_kernel<uint16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const uint16_t*)digits, signs, histA, P.n, sort_bins, sort_lb, slices, slice_len, (uint32_t)wb, wpg, row_i0, row_i1, blo, bcnt);
    else sortA_hist_kernel<int16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const int16_t*)digits, signs, histA, P.n, sort_bins, sort_lb, slices, slice_len, (uint32_t)wb, wpg, row_i0, row_i1, blo, bcnt);
    sortA_colscan_kernel<<<dim3((sort_bins + SORT_THREADS / 64 - 1) / (SORT_THREADS / 64), G), SORT_THREADS, 0, st>>>(histA, bintot, sort_bins, slices);
    sortA_scan_kernel<<<G, SORT_THREADS, 0, st>>>(bintot, binstart, totals, sort_bins, G == 1 ? bases : nullptr, gstart + (size_t)G * P.nb);
    if (G > 1) sort_bases_kernel<<<1, 64, 0, st>>>(totals, bases, gstart + (size_t)G * P.nb, G);
    static const int staged = [] { const char* e = getenv("BBGPU_SORT_STAGED"); return e ? atoi(e) : 3; }(); // tuning knob: bit 0 pass B, bit 1 pass A
    if ((staged & 2) && (P.n & 7u) == 0) {
        if (wide) sortA_scatter_staged_kernel<uint16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const uint16_t*)digits, signs, histA, binstart, bases, tmp_entries, P.n, sort_bins, sort_lb, slices,
                                                                                  slice_len, (uint32_t)wb, wpg, idx_stride, P.W, row_i0, row_i1, blo, bcnt);
        else sortA_scatter_staged_kernel<int16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const int16_t*)digits, signs, histA, binstart, bases, tmp_entries, P.n, sort_bins, sort_lb, slices,
                                                                            slice_len, (uint32_t)wb, wpg, idx_stride, P.W, row_i0, row_i1, blo, bcnt);
    } else
    if (wide) sortA_scatter_kernel<uint16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const uint16_t*)digits, signs, histA, binstart, bases, tmp_entries, P.n, sort_bins, sort_lb, slices,
                                                                   slice_len, (uint32_t)wb, wpg, idx_stride, P.W, row_i0, row_i1, blo, bcnt);
    else sortA_scatter_kernel<int16_t><<<dim3(slices, G), SORT_THREADS, 0, st>>>((const int16_t*)digits, signs, histA, binstart, bases, tmp_entries, P.n, sort_bins, sort_lb, slices,
                                                                   slice_len, (uint32_t)wb, wpg, idx_stride, P.W, row_i0, row_i1, blo, bcnt);
    if (!(staged & 1)) sortB_kernel<<<dim3(sort_bins, G), table ? SORT_THREADS : 256, 0, st>>>(tmp_entries, binstart, bases, sorted, gstart, sort_bins, sort_lb, P.nb);
    else if (table) {
        // large MSMs: eight pieces per bin (one more launch, ~8 us on the front; small MSMs are chains of dependent launches and keep one)
        static const int parts_env = [] { const char* e = getenv("BBGPU_SORTB_PARTS"); return e ? std::min((int)SORTB_MAX_PARTS, std::max(1, atoi(e))) : 0; }(); // tuning knob
        uint32_t parts = parts_env ? (uint32_t)parts_env : (((uint64_t)n * nw1 >= ((uint64_t)1 << 21) && G <= MSM_MAX_JOBS) ? SORTB_MAX_PARTS : 1u);
        if ((size_t)G * sort_bins * parts * 128 * 4 > LY.histB_bytes) parts = 1; // the workspace keeps the piece counts only for large MSMs (carve)
        if (parts > 1) sortB_count_kernel<SORT_THREADS><<<dim3(sort_bins, G, parts), SORT_THREADS, 0, st>>>(tmp_entries, binstart, bases, histB, sort_bins, parts);
        sortB_staged_kernel<SORT_THREADS><<<dim3(sort_bins, G, parts), SORT_THREADS, 0, st>>>(tmp_entries, binstart, bases, sorted, gstart, sort_bins, sort_lb, P.nb, histB, parts);
    } else sortB_staged_kernel<256><<<dim3(sort_bins, G), 256, 0, st>>>(tmp_entries, binstart, bases, sorted, gstart, sort_bins, sort_lb, P.nb, nullptr, 1u);
    if (tm_acc) HIPCHK(hipEventRecord(ev[2], st));
    // K4 + K4m
    const uint32_t total_buckets = G * P.nb;
    // a bucket-range share expects its fraction of the entries (uniform digits); the grid below still covers the worst case
    const uint64_t m_expected = bshare ? std::max<uint64_t>(1, (uint64_t)n * nw * bcnt / P.nb) : (uint64_t)n * nw;
    const uint32_t merge_buckets = bshare ? bcnt : total_buckets; // buckets the merge and the folds visit
    // Other MSMs in flight (S.throughput): the chip is shared and what counts is the instructions this one issues.  Lanes with fewer than ~20 entries
    // pay their prologue (start-bucket search, first row) and their partial sums (one per lane and bucket touched: the merge's work) for little;
    // measured on 1/8 shares of a 2^20 MSM, four in flight: chunks of 10 (the latency choice) 0.202 ms per step, 15: 0.191, 20: 0.181, 25: 0.190, 30: 0.183;
    // 2^16 points, three in flight: 0.126 -> 0.118 ms (tools/point_share_ab.py, tools/msm_ab.py with BBGPU_CHUNK).
    static const int tp_env = [] { const char* e = getenv("BBGPU_THROUGHPUT"); return e ? atoi(e) : -1; }(); // tuning knob: 0 / 1 force the latency / throughput choices
    const bool tp = (tp_env < 0 ? hint : tp_env != 0) && jobs == 1;
    constexpr uint32_t TP_MIN_CHUNK = 20;
    uint32_t ch = bshare ? chunk_len_m(m_expected) : chunk_len(n, nw);
    ch = std::max(ch, min_chunk((uint64_t)n * nw)); // the workspace's partial slots are laid out for chunks no shorter than the WHOLE list's minimum (carve: chunks_cap)
    if (tp && ch < TP_MIN_CHUNK && !chunk_forced()) // ... as long as one workgroup per CU is left (a 2-of-17-window share of 2^16 points, 131 k entries: chunks of 20 0.091 ms per step, of 8 0.072)
        ch = std::max(ch, std::min<uint32_t>(TP_MIN_CHUNK, (uint32_t)(m_expected / (acc_capacity_lanes() / acc_wg_per_cu()))));
    // merge: 2^logG lanes per bucket, sized for the expected number of partials per bucket (~ entries / (buckets * ch) + 1);
    // a bucket cut into more than 8 partials per lane of its group is queued for the workgroup-per-bucket kernel
    // -- but no wider than what fills the chip once (~2^16 lanes): beyond that the extra lanes only add issue work
    const uint32_t avg_partials = (uint32_t)(m_expected / ((uint64_t)merge_buckets * ch)) + 1;
    uint32_t logG = 0;
    static const uint32_t lanes_log = [] { const char* e = getenv("BBGPU_MERGE_LANES_LOG"); return e ? (uint32_t)std::min(20, std::max(14, atoi(e))) : 16u; }(); // tuning knob
    while ((1u << logG) < avg_partials && logG < 6 && ((uint64_t)merge_buckets << (logG + 1)) <= ((uint64_t)1 << lanes_log)) logG++;
    const uint32_t merge_light = std::max(6u, 8u << logG);
    const uint32_t max_chunks = (uint32_t)(((uint64_t)n * nw + ch - 1) / ch);
    // An MSM is a chain of dependent launches: its three helper launches -- two fills and the device-to-host copy, ~5 us each -- are folded
    // into the kernels around them, the last kernel writing the 8 KiB of results straight into the pinned host buffer.  (Round 1 measured the
    // folded tail 1..3 % SLOWER per pipelined 2^20 step and kept it for small MSMs only; with the round-2 tail it is level or ahead at every
    // size -- 1.371 vs 1.385 ms latency, 1.174 vs 1.176 ms per step, tools/msm_ab.py, two alternating runs in one box -- and is the one path.)
    static const int fold_env = [] { const char* e = getenv("BBGPU_FOLD_TAIL"); return e ? atoi(e) : 1; }(); // tuning knob: 0 = separate fills and copy
    const bool fold = fold_env != 0;
    msm_accumulate_kernel<<<(max_chunks + MSM_THREADS - 1) / MSM_THREADS, MSM_THREADS, acc_lds_reserve(), st>>>(points, sorted, gstart, partials, total_buckets, ch, acc_prio(), fold ? heavy : nullptr);
    if (tm_acc) {
        HIPCHK(hipEventRecord(ev[3], st));
        if (int rc = acc_ring_record(S, st)) return rc;
    } else {
        S.acc_seq = 0;
    }
    if (!fold) HIPCHK(hipMemsetAsync(heavy, 0, HEAVY_IDS * 4, st));
    static const bool quad_tail = [] { const char* e = getenv("BBGPU_QUAD_TAIL"); return !e || atoi(e) != 0; }(); // 0: one point per lane (round 1)
    static const uint64_t quad_merge_max_entries = [] { const char* e = getenv("BBGPU_QUAD_MERGE_MAX_LOG"); return (uint64_t)1 << (e ? std::min(30, std::max(10, atoi(e))) : 21); }(); // tuning knob
    static const int quad_merge = [] { const char* e = getenv("BBGPU_QUAD_MERGE"); return e ? atoi(e) : 1; }(); // tuning knob: 0 off, 1 on, 2.. = 1 + forced logQ
    // A quad addition is 4 x 1,350 lane-instructions against 3,700 for one point per lane: quads shorten the chain where the lanes do not
    // fill the chip, and cost issue slots where they do.  2^20 points: 196,608 additions, ~42 us either way (one wave per SIMD and three
    // dependent 9-us additions, or four waves per SIMD sharing the multiplier) -- the plain kernel stays.  2^16 points x 17 windows
    // (8 partials per bucket): 2 quads per bucket 0.261 -> 0.253 ms latency; a 2-of-17-window share 0.200 -> 0.187 ms (tools/msm_ab.py).
    if (quad_tail && (quad_merge > 1 || (quad_merge == 1 && (uint64_t)n * nw <= quad_merge_max_entries))) {
        // quads per bucket: about half the expected number of partials, within one resident wave of tail workgroups (~2^18 lanes)
        uint32_t logQ = 0;
        while ((2u << logQ) < avg_partials && logQ < 4 && ((uint64_t)merge_buckets << (logQ + 3)) <= ((uint64_t)1 << 17)) logQ++;
        if (quad_merge > 1) logQ = std::min(4, quad_merge - 2);
        msm_merge_quad_kernel<<<(uint32_t)((((uint64_t)merge_buckets << (logQ + 2)) + MSM_THREADS - 1) / MSM_THREADS), MSM_THREADS, 0, st>>>(gstart, partials, buckets, heavy, blo, blo + merge_buckets, ch,
                                                                                                                                  std::max(32u, 8u << logQ), logQ);
    } else
    msm_merge_kernel<<<(uint32_t)((((uint64_t)merge_buckets << logG) + MSM_THREADS - 1) / MSM_THREADS), MSM_THREADS, 0, st>>>(gstart, partials, buckets, heavy,
                                                                                                                     blo, blo + merge_buckets, ch, merge_light, logG);
    msm_merge_heavy_kernel<<<HEAVY_WGS, MSM_THREADS, 0, st>>>(gstart, partials, buckets, heavy, ch, heavy + (((size_t)G * P.nb + HEAVY_IDS + 63) & ~(size_t)63));
    if (tm) HIPCHK(hipEventRecord(ev[4], st));

    // K5: bucket b = hi * 2^l + lo carries weight b + 1:
    //   S_w = Z + sum_lo lo * C_lo + 2^l * sum_hi hi * R_hi,   R = row sums (over lo), C = column sums (over hi), Z = sum R
    //   sum_hi hi * R_hi = sum_k 2^k TR_k, TR_k = sum of the R_hi whose bit k is set (same for C).
    const uint32_t H = 1u << P.hbits, L = 1u << P.lbits;
    {
        // row + column sums in one launch (one workgroup tree per row / column), then Z and the bit-sliced sums in a second
        uint32_t* Rr = scratch;
        uint32_t* Cc = scratch + (size_t)G * H * 32;
        uint32_t* const hout = (uint32_t*)ws.h_out + (size_t)PC.hout_group * 64 * 32; // this piece's groups of the pinned result array
        uint32_t* dest = fold ? hout : texp; // pinned host memory is device-accessible under the same pointer
        const uint32_t zero_words = G * 64 * 32;
        static const bool two_step_env = [] { const char* e = getenv("BBGPU_ROWCOL_TWO_STEP"); return !e || atoi(e) != 0; }(); // 0: the one-launch quad form for every size
        if (quad_tail && two_step_env && !bshare && rowcol_two_step(P) && (tp || jobs > 1)) { // alone, the one-launch form is ~20 us shorter
            uint32_t* segs = (uint32_t*)(p + LY.segs);
            const uint32_t seg_lanes = 2 * (H * L / ROWCOL_SEG);
            msm_rowcol_seg_kernel<<<dim3((seg_lanes + MSM_THREADS - 1) / MSM_THREADS, G), MSM_THREADS, 0, st>>>(buckets, segs, H, L, fold ? dest : nullptr, zero_words);
            msm_segsum_quad_kernel<<<dim3(H + L, G), 64, 0, st>>>(segs, Rr, Cc, H, L);
        } else
        if (quad_tail) msm_rowcol_quad_kernel<<<dim3((bshare ? brow1 - brow0 : H) + L, G), QFOLD_T, 0, st>>>(buckets, Rr, Cc, H, L, fold ? dest : nullptr, zero_words, bshare ? brow0 : 0u,
                                                                                                   bshare ? brow1 - brow0 : H);
        else msm_rowcol_kernel<<<dim3(H + L, G), std::max(H, L), 0, st>>>(buckets, Rr, Cc, H, L, fold ? dest : nullptr, zero_words);
        if (tm) HIPCHK(hipEventRecord(ev[5], st));
        if (!fold) HIPCHK(hipMemsetAsync(texp, 0, (size_t)G * 64 * 128, st)); // unused slots = infinity (zz = 0)
        if (quad_tail) msm_final_quad_kernel<<<dim3(1 + P.hbits + P.lbits, G), QFOLD_T, 0, st>>>(Rr, Cc, dest, P.hbits, P.lbits);
        else msm_final_kernel<<<dim3(1 + P.hbits + P.lbits, G), std::max(H, L), 0, st>>>(Rr, Cc, dest, P.hbits, P.lbits);
        if (tm) HIPCHK(hipEventRecord(ev[6], st));
    }
    if (!fold) HIPCHK(hipMemcpyAsync((uint32_t*)ws.h_out + (size_t)PC.hout_group * 64 * 32, texp, (size_t)G * 64 * 128, hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(S.done, st)); // re-recorded by every piece: the event of the last one covers them all (one stream)
    HIPCHK(launch_check());
    S.npieces++;
    S.pending = true;
    return BBGPU_OK;
}

// Waits for an issued MSM and finishes it on the host:
//   S_w = Z + sum_k 2^k TC_k + 2^l sum_k 2^k TR_k ;  result = sum_w 2^(c (wb + w)) S_w   (Horner from the top)
static host::Xyzz group_sum(const MsmSlot& S, const MsmPiece& PC, uint32_t w)
{
    auto pt = [&](uint32_t slot) {
        host::Xyzz q;
        memcpy(&q, (const uint8_t*)S.ws.h_out + ((size_t)(PC.hout_group + w) * 64 + slot) * 128, 128);
        return q;
    };
    host::Xyzz rs = host::g1_infinity();
    for (int k = (int)PC.hbits - 1; k >= 0; --k) rs = host::g1_add(host::g1_dbl(rs), pt(1 + k));
    for (uint32_t k = 0; k < PC.lbits; k++) rs = host::g1_dbl(rs);
    host::Xyzz cs = host::g1_infinity();
    for (int k = (int)PC.lbits - 1; k >= 0; --k) cs = host::g1_add(host::g1_dbl(cs), pt(32 + k));
    return host::g1_add(host::g1_add(rs, cs), pt(0));
}
static int finish_timing(MsmSlot& S, MsmTiming* timing)
{
    if (S.timed && timing) {
        float ms;
        timing->count = 0;
        if (S.timed_light) { // only the accumulation was bracketed
            for (int i = 0; i < 7; i++) timing->ms[timing->count++] = 0.0f;
            HIPCHK(hipEventElapsedTime(&ms, S.ev[2], S.ev[3]));
            timing->ms[3] = ms;
        } else {
            HIPCHK(hipEventElapsedTime(&ms, S.ev[0], S.ev[6]));
            timing->ms[timing->count++] = ms;
            for (int i = 0; i < 6; i++) {
                HIPCHK(hipEventElapsedTime(&ms, S.ev[i], S.ev[i + 1]));
                timing->ms[timing->count++] = ms;
            }
        }
        float exec = timing->ms[3];
        // previous timed accumulation still in the ring (not overwritten by a later issue) and already finished (it precedes this one)
        if (S.acc_seq > 1 && g_acc_seq - (S.acc_seq - 1) < (uint64_t)ACC_RING &&
            hipEventElapsedTime(&ms, g_acc_end[(S.acc_seq - 1) % ACC_RING], g_acc_end[S.acc_seq % ACC_RING]) == hipSuccess && ms > 0.0f && ms < exec)
            exec = ms;
        (void)hipGetLastError();
        timing->ms[timing->count++] = exec;
    }
    return BBGPU_OK;
}
int msm_finish(MsmSlot& S, host::Xyzz* result, MsmTiming* timing)
{
    *result = host::g1_infinity();
    if (!S.pending) return BBGPU_ERR_STATE;
    if (S.jobs > 1) return BBGPU_ERR_STATE; // a batch is collected with msm_finish_batch
    S.pending = false;
    bool waited = false;
    host::Xyzz total = host::g1_infinity();
    for (int pi = 0; pi < S.npieces; pi++) {
        const MsmPiece& PC = S.piece[pi];
        if (PC.trivial) continue;
        if (!waited) HIPCHK(hipEventSynchronize(S.done));
        waited = true;
        host::Xyzz acc = host::g1_infinity();
        for (int w = (int)PC.nw - 1; w >= 0; --w) {
            for (uint32_t k = 0; k < PC.c; k++) acc = host::g1_dbl(acc);
            acc = host::g1_add(acc, group_sum(S, PC, (uint32_t)w));
        }
        for (uint32_t k = 0; k < PC.c * PC.wb; k++) acc = host::g1_dbl(acc);
        total = host::g1_add(total, acc);
    }
    *result = total;
    if (!waited) return BBGPU_OK;
    return finish_timing(S, timing);
}
// one result per job of a batch (table mode: every job is one bucket set, no positional doublings)
int msm_finish_batch(MsmSlot& S, host::Xyzz* results, MsmTiming* timing)
{
    if (S.jobs <= 1) return msm_finish(S, results, timing); // a batch of one is an ordinary MSM (any mode)
    if (!S.pending) return BBGPU_ERR_STATE;
    S.pending = false;
    for (uint32_t j = 0; j < S.jobs; j++) results[j] = host::g1_infinity();
    bool waited = false;
    for (int pi = 0; pi < S.npieces; pi++) {
        const MsmPiece& PC = S.piece[pi];
        if (PC.trivial) continue;
        if (!waited) HIPCHK(hipEventSynchronize(S.done));
        waited = true;
        for (uint32_t j = 0; j < S.jobs; j++) results[j] = host::g1_add(results[j], group_sum(S, PC, j));
    }
    if (!waited) return BBGPU_OK;
    return finish_timing(S, timing);
}

void MsmSlot::release()
{
    ws.release();
    if (done) (void)hipEventDestroy(done);
    done = nullptr;
    for (auto& e : ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
    pending = false;
    npieces = 0;
    helper = -1;
    is_helper = append = false;
}

// ---- SRS management --------------------------------------------------------------------------------------------------
// stride_bytes: 128 = the reference's 2n-entry endomorphism table (base points at the even entries), 64 = a plain n-entry point table
int srs_upload(const uint64_t* host_table, size_t n, uint32_t** d_srs_out, hipStream_t st, size_t stride_bytes)
{
    DevBuf tab, srs; // freed on every error path (dev_free waits for the device: a kernel still reading them has finished by then)
    HIPCHK(dev_malloc(&tab.p, n * stride_bytes));
    HIPCHK(dev_malloc(&srs.p, n * 64));
    if (int rc = host_to_device(tab.p, host_table, n * stride_bytes, st)) return rc;
    srs_convert_kernel<<<(uint32_t)((n + 127) / 128), 128, 0, st>>>(tab.as<uint32_t>(), srs.as<uint32_t>(), (uint32_t)n, (uint32_t)(stride_bytes / 4));
    HIPCHK(launch_check());
    HIPCHK(hipStreamSynchronize(st));
    *d_srs_out = srs.release<uint32_t>();
    return BBGPU_OK;
}

// the same into buffers the caller owns (d_raw: n * stride_bytes, d_srs: n * 64), everything enqueued on `st`, nothing waited for: the caller runs its kernels
// behind it on the same stream (small tables that are used once: no allocation, no free, no synchronisation on their path)
int srs_upload_into(const uint64_t* host_table, size_t n, uint32_t* d_raw, uint32_t* d_srs, hipStream_t st, size_t stride_bytes)
{
    if (int rc = host_to_device(d_raw, host_table, n * stride_bytes, st)) return rc;
    srs_convert_kernel<<<(uint32_t)((n + 127) / 128), 128, 0, st>>>(d_raw, d_srs, (uint32_t)n, (uint32_t)(stride_bytes / 4));
    HIPCHK(launch_check());
    return BBGPU_OK;
}

// builds the pre-shifted window tables for a resident SRS of n points: W x n x 64 bytes
// windows [w_begin, w_end) only; *d_alloc_out is what dev_free takes, *d_tab_out the (virtual) address of window 0
int srs_build_table(const uint32_t* d_srs, size_t n, int c, int num_windows, int w_begin, int w_end, uint32_t** d_alloc_out, uint32_t** d_tab_out, hipStream_t st)
{
    DevBuf alloc;
    HIPCHK(dev_malloc(&alloc.p, (size_t)(w_end - w_begin) * n * 64));
    uint32_t* d_tab = alloc.as<uint32_t>() - (size_t)w_begin * n * 16;
    srs_table_kernel<<<(uint32_t)((n + MSM_THREADS - 1) / MSM_THREADS), MSM_THREADS, 0, st>>>(d_srs, d_tab, (uint32_t)n, make_layout(c, true), (uint32_t)num_windows,
                                                                                          (uint32_t)w_begin, (uint32_t)w_end);
    HIPCHK(launch_check());
    HIPCHK(hipStreamSynchronize(st));
    *d_alloc_out = alloc.release<uint32_t>();
    *d_tab_out = d_tab;
    return BBGPU_OK;
}

int srs_generate(const uint64_t* x_mont256, size_t first, size_t n, uint32_t** d_srs_out, uint64_t* host_table_out, hipStream_t st)
{
    DevBuf tab, srs;
    HIPCHK(dev_malloc(&tab.p, 32 * 256 * 128));
    HIPCHK(dev_malloc(&srs.p, n * 64));
    srs_gen_table_kernel<<<1, 32, 0, st>>>(tab.as<uint32_t>());
    // x: Montgomery 2^256 -> 2^261
    uint32_t w[8];
    for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)x_mont256[i]; w[2 * i + 1] = (uint32_t)(x_mont256[i] >> 32); }
    Fe<Fr, 1, 2> x261 = m256_to_m261<Fr>(unpack<Fr>(w));
    uint32_t cw[8];
    to_canonical(x261, cw);
    Fe<Fr, 1, 6> xc = unpack<Fr>(cw);
    Limbs9 xl;
    for (int i = 0; i < NL; i++) xl.d[i] = xc.d[i];
    srs_gen_points_kernel<<<(uint32_t)((n + 63) / 64), 64, 0, st>>>(tab.as<uint32_t>(), xl, srs.as<uint32_t>(), (uint32_t)n, (uint32_t)first);
    HIPCHK(launch_check());
    if (host_table_out) {
        DevBuf exp;
        HIPCHK(dev_malloc(&exp.p, n * 128));
        srs_export_kernel<<<(uint32_t)((n + 127) / 128), 128, 0, st>>>(srs.as<uint32_t>(), exp.as<uint32_t>(), (uint32_t)n);
        HIPCHK(launch_check());
        if (int rc = device_to_host_sync(host_table_out, exp.p, n * 128, st)) return rc;
    }
    HIPCHK(hipStreamSynchronize(st));
    *d_srs_out = srs.release<uint32_t>();
    return BBGPU_OK;
}

} // namespace bbgpu
