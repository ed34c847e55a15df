// ntt.hip -- radix-2 NTT / coset-FFT over BN254 Fr for gfx950.
//
// Replaces polynomial_arithmetic::fft / ifft / coset_fft / coset_ifft / *_with_constant
// (reference src/barretenberg/polynomials/polynomial_arithmetic.cpp:129-315, scale_by_generator :81-102).
// Contract kept: natural order in, natural order out, in place on n x 32-byte Montgomery(2^256) elements,
// inputs anywhere in [0, 2^256) (the prover hands [0,2p)), outputs canonical [0,p).
//
// Structure (MI355X-first, not the reference's log2(n) streaming rounds over a 2n-entry twiddle table):
//   n = n1 * n2, two HBM passes ("four-step"), each pass runs complete n1- / n2-point sub-transforms inside LDS:
//     pass 1: for every column j2: A[k1][j2] = sum_j1 x[j1*n2 + j2] w_n1^(j1 k1); then twist by w_n^(j2 k1)
//     pass 2: for every row k1:    X[k1 + n1 k2] = sum_j2 A[k1][j2] w_n2^(j2 k2)
//   so the vector crosses HBM twice (read+write) instead of ~log2(n) times, and twiddles are tiny tables
//   (n1/2 + n2/2 + 2*sqrt(n) entries) generated on the device -- the reference's 64 MiB table never exists.
//   Coset / inverse / constant scalings are fused into the first load and the last store.
//   Data stays in the reference's Montgomery(2^256) form end to end: only twiddles live in our 2^261 form, since
//   mont261(x * 2^256, w * 2^261) = x w 2^256.
// Field arithmetic: fe.hpp (9 x 29-bit limbs, lazy; ~163 v_mad_u64_u32 per multiply).  The kernels are bound by
// the integer VALU rate, not HBM: (n/2) log2 n + ~2n multiplies at ~1.4e11 mul/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unordered_map>

#include "bbgpu_internal.h"
#include "fe.hpp"

namespace bbgpu {

using Fr = FrP;
// Phase ablation for timing experiments (DESIGN_HISTORY 4): a BUILD variant like the JUNK knobs (make variant NAME=nttskip1 EXTRA=-DBBGPU_NTT_DEBUG_SKIP=1),
// never an environment variable -- the shipped kernels have no switch that changes results.  Bit 0: skip the stages, bit 1: skip the twist / scaling products.
#ifdef BBGPU_NTT_DEBUG_SKIP
constexpr uint32_t NTT_DEBUG_SKIP = BBGPU_NTT_DEBUG_SKIP;
#else
constexpr uint32_t NTT_DEBUG_SKIP = 0;
#endif
constexpr int NTT_VMAX = 48;                 // lazy value bound inside one pass: 6 + 3 * 12 stages + slack
using FrL = Fe<Fr, 1, NTT_VMAX>;             // LDS-resident element
// The fused pass keeps LAZIER limbs in LDS (round 3): up to 4 U.  A radix-2^2 group then needs two renormalisations instead of four -- the
// element in the x0 role (never multiplied inside the group) when it is loaded, and the one output whose bound would reach 5 U -- because
// the other three inputs go through the multiplier (L1 L2 <= 6) and differences with a PRODUCT as subtrahend need the small borrow-proofing
// offset (fe.hpp FeE): 24 instructions each, 12 of them saved per group and pass at log_s = 10 (-4.6 % of the pass's VALU instructions).
constexpr int NTT_LDSL = 4;
using FrS = Fe<Fr, NTT_LDSL, NTT_VMAX>;
constexpr int NTT_MAX_LOG2N = 28;            // two-adicity of r - 1 (fr.hpp:60-63): the largest domain the reference has a root for
constexpr int NTT_MAX_LOG_SUB = 11;          // sub-transform up to 2048 points (72 KiB of LDS)
constexpr int NTT_LDS_ELEMS = 2048;          // elements of LDS per workgroup (9 words each = 72 KiB) -> 2 WG / CU
constexpr int NTT_THREADS = 512;
constexpr int NTT_FULL_TWIST_MAX_LOG2N = 22; // 2 x n x 32 B of twist factors per domain: 256 MiB at 2^22

// proof obligation is the caller's: the true bounds are <= (L, V)
template <int L, int V, class F, int L2, int V2> BB_HD Fe<F, L, V> assume_bound(const Fe<F, L2, V2>& a)
{
    Fe<F, L, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i];
    return r;
}

__device__ __forceinline__ void load8(const uint32_t* p, uint32_t (&w)[8])
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store8(uint32_t* p, const uint32_t (&w)[8])
{
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// table entries are canonical Montgomery-261 values stored PRE-UNPACKED: the nine 29-bit limbs the multiplier takes, padded to
// 12 words (three 16-byte loads, no shift/mask work per use: the packed 8-word form cost ~20 VALU instructions per twiddle, three
// twiddles per radix-2^2 group).  The tables are tiny (n1/2 + n2/2 + 4 sqrt(n) entries), so 48 instead of 32 bytes per entry is free.
constexpr int TW_WORDS = 12;
__device__ __forceinline__ FeT<Fr> load_tw(const uint32_t* table, uint32_t idx)
{
#ifdef BBGPU_NTT_ABL_TW // timing ablation (WRONG results; a build variant like BBGPU_NTT_DEBUG_SKIP): every lane reads entry (idx & mask) -- 0: one line per load
    idx &= BBGPU_NTT_ABL_TW;
#endif
    const uint4* q = reinterpret_cast<const uint4*>(table + TW_WORDS * (size_t)idx);
    const uint4 a = q[0], b = q[1], c = q[2];
    FeT<Fr> r;
    r.d[0] = a.x; r.d[1] = a.y; r.d[2] = a.z; r.d[3] = a.w;
    r.d[4] = b.x; r.d[5] = b.y; r.d[6] = b.z; r.d[7] = b.w;
    r.d[8] = c.x;
    return r;
}

// the same entry out of LDS (north star: "LDS-staged twiddles"; fused kernel, 2048-element tiles): 12-word entries, three 16-byte reads
__device__ __forceinline__ FeT<Fr> load_tw_lds(const uint32_t* lds_table, uint32_t idx)
{
    const uint4* q = reinterpret_cast<const uint4*>(lds_table + TW_WORDS * idx);
    const uint4 a = q[0], b = q[1], c = q[2];
    FeT<Fr> r;
    r.d[0] = a.x; r.d[1] = a.y; r.d[2] = a.z; r.d[3] = a.w;
    r.d[4] = b.x; r.d[5] = b.y; r.d[6] = b.z; r.d[7] = b.w;
    r.d[8] = c.x;
    return r;
}
// Entries of the sub-transform's twiddle table the middle stage pairs up to s = 6 read: indices j << (log_s - 1 - s), j << (log_s - 2 - s), (j + m) << (log_s - 2 - s),
// j < m = 2^s -- all multiples of 2^(log_s - 8), at most 127 of them: 128 entries = 6 KiB staged beside a 72 KiB tile (two workgroups per CU: 159,744 of 163,840 bytes).
constexpr uint32_t NTT_TW_LDS_ENTRIES = 128;

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits)
{
    return __brev(x) >> (32 - bits);
}
// LDS position of element e of a column: XOR swizzle of the five bank bits.  A bijection on [0, 2^k) for every k, found
// by search: every access pattern of the radix-2^2 stage pairs (strides 4, 16, 64, ...), of the odd last stage and of
// the natural-order store is bank-conflict free, and the bit-reversed load is <= 2-way (4 at 2048); the unswizzled
// layout is 4-, 4-, 2-way conflicted in the first three stage pairs and 32-way in the load.
__device__ __forceinline__ uint32_t lds_pos(uint32_t e)
{
    return e ^ ((e >> 2) & 31u) ^ ((e >> 7) & 3u);
}

struct NttPassArgs {
    const uint32_t* in;       // n x 8 words
    uint32_t* out;            // n x 8 words
    const uint32_t* tw_sub;   // w_S^k, k < S/2
    const uint32_t* twist_lo; // w_n^l, l < 2^lo_bits          (pass 1 only)
    const uint32_t* twist_hi; // w_n^(h << lo_bits)
    const uint32_t* row_scale;  // FLAGS & 64: factor per INPUT row a of the sub-transform (coset pre-scale), FLAGS & 128: per OUTPUT index k (coset post-scale); 12-word entries
    const uint32_t* twist_full; // FLAGS & 32: w_n^(b k) for every element of pass 1's output, packed 8 words, in the output's own layout
    const uint32_t* scale_lo; // g^l (pre) or g^-l (post) two-level tables, lo part
    const uint32_t* scale_hi;
    uint32_t post_const[NL];  // Montgomery-261 constant applied to every output of the last pass
    uint32_t log_s;           // sub-transform size S = 2^log_s
    uint32_t log_b;           // number of sub-transforms B = 2^log_b   (n = S * B)
    uint32_t cols, log_cols;  // sub-transforms per workgroup (power of two, cols * S <= NTT_LDS_ELEMS)
    uint32_t in_sa, in_sb;    // element (a, b) is read from in[a * in_sa + b * in_sb]
    uint32_t out_sa, out_sb;  // result (k, b) is written to out[k * out_sa + b * out_sb]
    uint32_t lo_bits;         // split of the two-level tables
    uint32_t b_fast;          // 1: consecutive threads walk b first (column pass), 0: a first (row pass)
    uint32_t half_tile;       // host: launch the 256-thread instance (tiles of 1024 elements)
    uint32_t store_b_fast;    // fused kernel: the same choice for the STORE (1 when the b index is the contiguous one on the output side)
    uint32_t xcd_remap;       // 1: contiguous tile range per XCD (see ntt_pass_kernel)
    uint32_t batch;           // transforms in this launch (blockIdx.y): transform j works on in + j * in_bstride -> out + j * out_bstride
    size_t in_bstride, out_bstride; // in words
    uint32_t tw_lds;          // fused kernel, 512-thread instance: stage the middle pairs' twiddles in LDS (the launch reserved NTT_TW_LDS_ENTRIES entries behind the tile)
    uint32_t nat_bstep;       // natural output index of batch item j starts at j * nat_bstep (three-pass transforms: rows of one big transform; else 0)
};

// FLAGS: 1 = pre-scale input by scale tables (coset_fft), 2 = twist output (pass 1 of 2),
//        4 = post-scale by scale tables (coset_ifft), 8 = post-scale by constant, 16 = emit canonical output
//        fused kernel only: 32 = the twist comes from a full-size table (one multiplication), 64 = pre-scale by a per-row table,
//        128 = post-scale by a per-output-index table.  64 / 128 are how the coset variants cost ONE extra multiplication per element
//        instead of two: g^(j1 n2 + j2) = (g^n2)^j1 * g^j2 -- the first factor is a row table of the pass-1 sub-transform, the second is
//        constant along a column and is folded into the (coset) twist table; likewise g^-(k1 + n1 k2) n^-1 on the way out.
template <int FLAGS> __global__ void __launch_bounds__(NTT_THREADS) ntt_pass_kernel(NttPassArgs A)
{
    extern __shared__ uint32_t lds[]; // [9][cols * S]
    const uint32_t S = 1u << A.log_s, cols = A.cols, E = cols * S;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive blockIdx would put
    // the neighbouring column tiles -- which share 128-byte lines, a tile row being only cols * 32 B wide -- on different L2s.
    // Give every XCD a contiguous range of tiles instead.
    uint32_t bid = blockIdx.x;
    if (A.xcd_remap && (gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t b0 = bid * cols;
    const uint32_t tid = threadIdx.x;

    // ---- load (+ optional coset pre-scale), bit-reversed into LDS ------------------------------------------------
    for (uint32_t e = tid; e < E; e += NTT_THREADS) {
        uint32_t a, c;
        if (A.b_fast) { c = e & (cols - 1); a = e >> A.log_cols; } else { a = e & (S - 1); c = e >> A.log_s; }
        const size_t gidx = (size_t)a * A.in_sa + (size_t)(b0 + c) * A.in_sb;
        uint32_t w[8];
        load8(A.in + (size_t)blockIdx.y * A.in_bstride + 8 * gidx, w);
        FrL x = unpack<Fr>(w);
        if constexpr (FLAGS & 1) {
            const uint32_t i = (uint32_t)gidx; // natural coefficient index
            auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
            x = mul(x, g);
        }
        const uint32_t pos = c * S + lds_pos(bitrev(a, A.log_s));
#pragma unroll
        for (int l = 0; l < NL; l++) lds[l * E + pos] = x.d[l];
    }
    __syncthreads();

    // ---- log_s radix-2 DIT stages in LDS, two stages per LDS round trip (radix-2^2 groups of 4 elements) -------------
    // Stage pair (s, s+1), m = 2^s: the group {e0, e0+m, e0+2m, e0+3m} is closed under both stages:
    //   stage s   : (e0,e1) and (e2,e3) with w_{2m}^j         stage s+1 : (e0,e2) with w_{4m}^j, (e1,e3) with w_{4m}^(j+m)
    // Intermediate sums stay lazy (no renormalisation between the two stages); same multiplies as radix-2, half the LDS
    // traffic and half the barriers.
    const uint32_t half = S >> 1;
    uint32_t s = (NTT_DEBUG_SKIP & 1) ? A.log_s : 0;
    for (; s + 1 < A.log_s; s += 2) {
        const uint32_t m = 1u << s, quarter = S >> 2, ngr = cols * quarter;
        for (uint32_t gq = tid; gq < ngr; gq += NTT_THREADS) {
            const uint32_t c = gq >> (A.log_s - 2), q = gq & (quarter - 1);
            const uint32_t j = q & (m - 1);
            const uint32_t i0 = ((q >> s) << (s + 2)) | j, cb = c * S;
            const uint32_t e0 = cb + lds_pos(i0), e1 = cb + lds_pos(i0 + m), e2 = cb + lds_pos(i0 + 2 * m), e3 = cb + lds_pos(i0 + 3 * m);
            FrL x0, x1, x2, x3;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                x0.d[l] = lds[l * E + e0];
                x1.d[l] = lds[l * E + e1];
                x2.d[l] = lds[l * E + e2];
                x3.d[l] = lds[l * E + e3];
            }
            FrL y0, y1, y2, y3;
            if (s == 0) { // first pair: stage 0 has twiddle one everywhere, stage 1 has twiddle one on (e0, e2): ONE multiplication for the group
                // value bound: inputs come straight from unpack()/pre-scale (< 6p), so the four sums stay far below NTT_VMAX
                using In = Fe<Fr, 1, 6>; // what unpack() / the pre-scale product really hold
                const In z0 = assume_bound<1, 6>(x0), z1 = assume_bound<1, 6>(x1), z2 = assume_bound<1, 6>(x2), z3 = assume_bound<1, 6>(x3);
                const auto a0 = add(z0, z1);
                const auto a1 = sub(z0, z1);
                const auto a2 = add(z2, z3);
                const auto a3 = sub(z2, z3);
                const FeT<Fr> w4 = load_tw(A.tw_sub, 1u << (A.log_s - 2)); // w_S^(S/4)
                const auto u3 = mul(w4, a3);                               // < 3p
                y0 = assume_bound<1, NTT_VMAX>(weak(add(a0, a2)));
                y2 = assume_bound<1, NTT_VMAX>(weak(sub(a0, a2)));
                y1 = assume_bound<1, NTT_VMAX>(weak(add(a1, u3)));
                y3 = assume_bound<1, NTT_VMAX>(weak(sub(a1, u3)));
            } else {
                // stage s
                const FeT<Fr> w1 = load_tw(A.tw_sub, j << (A.log_s - 1 - s));
                const auto t1 = mul(w1, x1), t3 = mul(w1, x3); // < 3p
                const auto a0 = add(x0, t1);
                const auto a1 = sub(x0, t1);
                const auto a2 = add(x2, t3);
                const auto a3 = sub(x2, t3);
                // stage s+1
                const FeT<Fr> w2a = load_tw(A.tw_sub, j << (A.log_s - 2 - s));
                const FeT<Fr> w2b = load_tw(A.tw_sub, (j + m) << (A.log_s - 2 - s));
                const auto u2 = mul(w2a, a2), u3 = mul(w2b, a3); // < 3p
                y0 = assume_bound<1, NTT_VMAX>(weak(add(a0, u2)));
                y2 = assume_bound<1, NTT_VMAX>(weak(sub(a0, u2)));
                y1 = assume_bound<1, NTT_VMAX>(weak(add(a1, u3)));
                y3 = assume_bound<1, NTT_VMAX>(weak(sub(a1, u3)));
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                lds[l * E + e0] = y0.d[l];
                lds[l * E + e1] = y1.d[l];
                lds[l * E + e2] = y2.d[l];
                lds[l * E + e3] = y3.d[l];
            }
        }
        __syncthreads();
    }
    if (s < A.log_s) { // odd number of stages: one plain radix-2 stage left
        const uint32_t m = 1u << s, nbf = cols * half;
        for (uint32_t bf = tid; bf < nbf; bf += NTT_THREADS) {
            const uint32_t c = bf >> (A.log_s - 1), i = bf & (half - 1);
            const uint32_t j = i & (m - 1);
            const uint32_t il = ((i >> s) << (s + 1)) | j;
            const uint32_t lo = c * S + lds_pos(il), hi = c * S + lds_pos(il + m);
            FrL x, y;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                x.d[l] = lds[l * E + lo];
                y.d[l] = lds[l * E + hi];
            }
            FrL xs, ys;
            if (s == 0) {
                xs = assume_bound<1, NTT_VMAX>(weak(add(x, y)));
                ys = assume_bound<1, NTT_VMAX>(weak(sub(x, y)));
            } else {
                const FeT<Fr> w = load_tw(A.tw_sub, j << (A.log_s - 1 - s));
                const auto t = mul(w, y);
                xs = assume_bound<1, NTT_VMAX>(weak(add(x, t)));
                ys = assume_bound<1, NTT_VMAX>(weak(sub(x, t)));
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                lds[l * E + lo] = xs.d[l];
                lds[l * E + hi] = ys.d[l];
            }
        }
        __syncthreads();
    }

    // ---- store (+ twist / post-scale / canonicalise) ---------------------------------------------------------------
    for (uint32_t e = tid; e < E; e += NTT_THREADS) {
        uint32_t k, c;
        if (A.b_fast) { c = e & (cols - 1); k = e >> A.log_cols; } else { k = e & (S - 1); c = e >> A.log_s; }
        const uint32_t b = b0 + c;
        FrL x;
#pragma unroll
        for (int l = 0; l < NL; l++) x.d[l] = lds[l * E + c * S + lds_pos(k)];
        const size_t gidx = (size_t)k * A.out_sa + (size_t)b * A.out_sb;
        uint32_t w[8];
        if ((NTT_DEBUG_SKIP & 2) != 0) {
            pack(assume_bound<1, 2>(x), w);
        } else if constexpr (FLAGS & 2) {
            const uint32_t ex = b * k; // < n
            auto tw = mul(load_tw(A.twist_lo, ex & ((1u << A.lo_bits) - 1)), load_tw(A.twist_hi, ex >> A.lo_bits));
            auto r = mul(x, tw); // 48 * 2 / 169 + 2 = 2  -> fits 256 bits
            pack(r, w);
        } else {
            Fe<Fr, 1, 3> r;
            if constexpr ((FLAGS & 4) && (FLAGS & 8)) {
                const uint32_t i = (uint32_t)gidx + blockIdx.y * A.nat_bstep;
                auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
                r = mul(mul(x, g), fe_from<Fr>(A.post_const));
            } else if constexpr (FLAGS & 4) {
                const uint32_t i = (uint32_t)gidx + blockIdx.y * A.nat_bstep;
                auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
                r = mul(x, g);
            } else if constexpr (FLAGS & 8) {
                r = mul(x, fe_from<Fr>(A.post_const));
            } else {
                r = reduce_value(x);
            }
            to_canonical(r, w);
        }
        store8(A.out + (size_t)blockIdx.y * A.out_bstride + 8 * gidx, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same pass with its first and last LDS round trips removed (sub-transforms of 16 points and more):
//   A  load + FIRST stage pair: a thread loads the four rows t, t + S/4, t + S/2, t + 3S/4 of a column (consecutive threads take
//      consecutive rows / columns: coalesced), runs the radix-2^2 group they form after bit reversal -- group bitrev(t) -- in registers
//      and only then writes LDS: no separate load phase, the butterflies of one wave run under the loads of the next;
//   B  the middle stage pairs in LDS as before;
//   C  LAST stage pair (or the odd last stage) + store: its four outputs j, j + S/4, j + S/2, j + 3S/4 are natural-order rows, so they go
//      through twist / scaling / canonicalisation straight to memory.
// Two barriers and two LDS round trips fewer per pass, and the memory phases are no longer pure waiting (all workgroups of a 2^20
// transform run in lockstep -- one resident wave of them holds the whole vector -- so nothing else could overlap them).
template <int FLAGS, int L> __device__ __forceinline__ void ntt_finish_store(const NttPassArgs& A, const Fe<Fr, L, NTT_VMAX>& x, uint32_t k, uint32_t b)
{
    static_assert(L <= 6, "the twist / scaling product takes limbs up to 6 U");
    const size_t gidx = (size_t)k * A.out_sa + (size_t)b * A.out_sb;
    uint32_t w[8];
    if ((NTT_DEBUG_SKIP & 2) != 0) {
        pack(assume_bound<L, 2>(x), w);
    } else if constexpr ((FLAGS & 2) && (FLAGS & 32)) {
        // one multiplication per element: the twist factor comes from a table as large as the vector, read exactly like the output is written
        uint32_t tw8[8];
        load8(A.twist_full + 8 * gidx, tw8);
        const FeT<Fr> tw = assume_bound<1, 1>(unpack<Fr>(tw8));
        pack_exact(mul(x, tw), w); // a product: limbs exact, value < 2p < 2^256 -- no carry chain
    } else if constexpr (FLAGS & 2) {
        const uint32_t ex = b * k; // < n
        auto tw = mul(load_tw(A.twist_lo, ex & ((1u << A.lo_bits) - 1)), load_tw(A.twist_hi, ex >> A.lo_bits));
        pack_exact(mul(x, tw), w); // 48 * 2 / 169 + 2 = 2  -> fits 256 bits; a product's limbs are exact
    } else {
        Fe<Fr, 1, 3> r;
        if constexpr (FLAGS & 128) {
            r = mul(x, load_tw(A.row_scale, k));
        } else if constexpr ((FLAGS & 4) && (FLAGS & 8)) {
            const uint32_t i = (uint32_t)gidx + blockIdx.y * A.nat_bstep;
            auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
            r = mul(mul(x, g), fe_from<Fr>(A.post_const));
        } else if constexpr (FLAGS & 4) {
            const uint32_t i = (uint32_t)gidx + blockIdx.y * A.nat_bstep;
            auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
            r = mul(x, g);
        } else if constexpr (FLAGS & 8) {
            r = mul(x, fe_from<Fr>(A.post_const));
        }
        if constexpr ((FLAGS & (128 | 4 | 8)) != 0) to_canonical(exact_limbs(r), w); // r is a product in every one of these branches
        else to_canonical(x, w); // plain transform: value reduction, ONE carry chain, two conditional subtractions
    }
    store8(A.out + (size_t)blockIdx.y * A.out_bstride + 8 * gidx, w);
}

// One radix-2^2 group (stages s, s + 1) on LDS-lazy inputs: x0 renormalised (nobody multiplies it), everything else straight into the
// multiplier; outputs at 3 / 4 / 4 / 5 U.  Limb bounds are checked by the types (mul's static_assert), value bounds as in the rest of the pass.
struct Radix4Out {
    Fe<Fr, 3, NTT_VMAX> y0;
    Fe<Fr, 4, NTT_VMAX> y1, y2;
    Fe<Fr, 5, NTT_VMAX> y3;
};
__device__ __forceinline__ Radix4Out radix4_lazy(const FrS& x0, const FrS& x1, const FrS& x2, const FrS& x3, const FeT<Fr>& w1, const FeT<Fr>& w2a, const FeT<Fr>& w2b)
{
    const FrL x0w = weak(x0);
    const auto t1 = exact_limbs(mul(w1, x1)), t3 = exact_limbs(mul(w1, x3)); // < 3p
    const auto a0 = add(x0w, t1);                                               // 2 U
    const auto a1 = sub(x0w, t1);                                               // 3 U
    const auto a2 = add(x2, t3);                                                // 5 U
    const auto a3 = sub(x2, t3);                                                // 6 U: the multiplier's limit
    const auto u2 = exact_limbs(mul(w2a, a2)), u3 = exact_limbs(mul(w2b, a3));
#ifdef BBGPU_NTT_JUNK // issue-model experiment (DESIGN_HISTORY 4): k extra cheap VALU instructions per multiplication of a stage pair, results unused
    {
        uint32_t j0 = x0.d[0], j1 = x1.d[0];
#pragma unroll
        for (int q = 0; q < 2 * BBGPU_NTT_JUNK; q++) asm volatile("v_and_b32 %0, 0x1fffffff, %1\n\tv_add_u32 %1, %0, %1" : "+v"(j0), "+v"(j1));
    }
#endif
    Radix4Out o;
    o.y0 = assume_bound<3, NTT_VMAX>(add(a0, u2));
    o.y2 = assume_bound<4, NTT_VMAX>(sub(a0, u2));
    o.y1 = assume_bound<4, NTT_VMAX>(add(a1, u3));
    o.y3 = assume_bound<5, NTT_VMAX>(sub(a1, u3));
    return o;
}

// orders a wave's LDS accesses among themselves (no instruction: the LDS runs one wave's instructions in issue order; the fences stop the compiler)
__device__ __forceinline__ void ntt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef NTT_OCC_ATTR
#define NTT_OCC_ATTR // A/B knob: e.g. -DNTT_OCC_ATTR='__attribute__((amdgpu_waves_per_eu(5,5)))'
#endif
template <int FLAGS, int THREADS> __global__ void __launch_bounds__(THREADS) NTT_OCC_ATTR ntt_pass_fused_kernel(NttPassArgs A)
{
    extern __shared__ uint32_t lds[]; // [9][E]
    // The plane stride is the TILE size of this instantiation (four elements per thread), a compile-time constant: the nine plane offsets of every LDS
    // access then sit in the instruction's offset field instead of costing a v_add each (round 4: ~30 of a stage pair's 1,049 VALU instructions);
    // a launch whose sub-transforms fill less than a tile (n below the tile size) uses the front of every plane.
#ifdef BBGPU_NTT_RUNTIME_STRIDE
    const uint32_t E = A.cols << A.log_s;
#else
    constexpr uint32_t E = (uint32_t)THREADS * 4u;
#endif
    const uint32_t S = 1u << A.log_s, cols = A.cols, quarter = S >> 2, ngr = cols * quarter;
    uint32_t bid = blockIdx.x;
    if (A.xcd_remap && (gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t b0 = bid * cols;
    const uint32_t tid = threadIdx.x;
    using In = Fe<Fr, 1, 6>; // what unpack() / the pre-scale product really hold
#ifndef BBGPU_NTT_TW_GLOBAL // (-DBBGPU_NTT_TW_GLOBAL: every twiddle from the L1 / L2-resident table, the round-2 .. 4 form: A/B)
    // LDS-staged twiddles of the middle stage pairs (A.tw_lds: the host asks for it where the launch reserved the room): entry e = table entry e << (log_s - 8),
    // copied by the first 384 threads while everybody's first loads are in flight; the barrier behind phase A publishes it
    uint32_t* const lds_tw = lds + NL * E;
    const uint32_t tw_shift = A.log_s - 8;
    if (A.tw_lds && tid < NTT_TW_LDS_ENTRIES * 3) {
        const uint32_t e = tid / 3, part = tid - 3 * e;
        reinterpret_cast<uint4*>(lds_tw + TW_WORDS * e)[part] = reinterpret_cast<const uint4*>(A.tw_sub + TW_WORDS * ((size_t)e << tw_shift))[part];
    }
#endif

    // ---- A: load (+ coset pre-scale) + stage pair (0, 1) ------------------------------------------------------------------
    for (uint32_t gq = tid; gq < ngr; gq += THREADS) {
        uint32_t c, t;
        if (A.b_fast) { c = gq & (cols - 1); t = gq >> A.log_cols; } else { t = gq & (quarter - 1); c = gq >> (A.log_s - 2); }
        In x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const size_t gidx = (size_t)(t + j * quarter) * A.in_sa + (size_t)(b0 + c) * A.in_sb;
            uint32_t w[8];
            load8(A.in + (size_t)blockIdx.y * A.in_bstride + 8 * gidx, w);
            x[j] = unpack<Fr>(w);
            if constexpr (FLAGS & 64) {
                x[j] = mul(x[j], load_tw(A.row_scale, t + j * quarter));
            } else if constexpr (FLAGS & 1) {
                const uint32_t i = (uint32_t)gidx; // natural coefficient index
                auto g = mul(load_tw(A.scale_lo, i & ((1u << A.lo_bits) - 1)), load_tw(A.scale_hi, i >> A.lo_bits));
                x[j] = mul(x[j], g);
            }
        }
        // row t + j S/4 sits at bit-reversed index 4 bitrev(t) + bitrev2(j): the group of LDS indices 4q .. 4q + 3 is rows (0, 2, 1, 3)
        const uint32_t q = bitrev(t, A.log_s - 2), cb = c * S;
        const auto z0 = exact_limbs(x[0]), z1 = exact_limbs(x[2]), z2 = exact_limbs(x[1]), z3 = exact_limbs(x[3]); // unpack() / products: exact limbs
        FrS y0, y1, y2, y3;
        if (NTT_DEBUG_SKIP & 1) {
            y0 = z0; y1 = z1; y2 = z2; y3 = z3;
        } else {
            const auto a0 = add(z0, z1);                                   // 2 U
            const auto a1 = sub(z0, z1);                                   // 3 U
            const auto a2 = add(z2, z3);                                   // 2 U
            const auto a3 = sub(z2, z3);                                   // 3 U
            const FeT<Fr> w4 = load_tw(A.tw_sub, 1u << (A.log_s - 2));     // w_S^(S/4)
            const auto u3 = exact_limbs(mul(w4, a3));                      // < 3p
            y0 = assume_bound<4, NTT_VMAX>(add(a0, a2));                   // 4 U: stays lazy
            y2 = assume_bound<1, NTT_VMAX>(weak(sub(a0, a2)));             // 6 U -> renormalised
            y1 = assume_bound<4, NTT_VMAX>(add(a1, u3));                   // 4 U
            y3 = assume_bound<1, NTT_VMAX>(weak(sub(a1, u3)));             // 5 U -> renormalised
        }
        const uint32_t e0 = cb + lds_pos(4 * q), e1 = cb + lds_pos(4 * q + 1), e2 = cb + lds_pos(4 * q + 2), e3 = cb + lds_pos(4 * q + 3);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            lds[l * E + e0] = y0.d[l];
            lds[l * E + e1] = y1.d[l];
            lds[l * E + e2] = y2.d[l];
            lds[l * E + e3] = y3.d[l];
        }
    }
    __syncthreads();

    // ---- B: middle stage pairs in LDS ----------------------------------------------------------------------------------------
    const bool odd = (A.log_s & 1) != 0;
    const uint32_t last_s = odd ? A.log_s - 1 : A.log_s - 2; // first stage of the part fused with the store
    uint32_t s = (NTT_DEBUG_SKIP & 1) ? last_s : 2;
    for (; s < last_s; s += 2) {
        const uint32_t m = 1u << s;
        for (uint32_t gq = tid; gq < ngr; gq += THREADS) {
            const uint32_t c = gq >> (A.log_s - 2), q = gq & (quarter - 1);
            const uint32_t j = q & (m - 1);
            const uint32_t i0 = ((q >> s) << (s + 2)) | j, cb = c * S;
            const uint32_t e0 = cb + lds_pos(i0), e1 = cb + lds_pos(i0 + m), e2 = cb + lds_pos(i0 + 2 * m), e3 = cb + lds_pos(i0 + 3 * m);
            FrS x0, x1, x2, x3;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                x0.d[l] = lds[l * E + e0];
                x1.d[l] = lds[l * E + e1];
                x2.d[l] = lds[l * E + e2];
                x3.d[l] = lds[l * E + e3];
            }
#ifndef BBGPU_NTT_TW_GLOBAL
            FeT<Fr> w1, w2a, w2b;
            if (A.tw_lds && s <= 6) { // (j << (log_s - 1 - s)) >> (log_s - 8) = j << (7 - s), ...
                w1 = load_tw_lds(lds_tw, j << (7 - s));
                w2a = load_tw_lds(lds_tw, j << (6 - s));
                w2b = load_tw_lds(lds_tw, (j + m) << (6 - s));
            } else {
                w1 = load_tw(A.tw_sub, j << (A.log_s - 1 - s));
                w2a = load_tw(A.tw_sub, j << (A.log_s - 2 - s));
                w2b = load_tw(A.tw_sub, (j + m) << (A.log_s - 2 - s));
            }
#else
            const FeT<Fr> w1 = load_tw(A.tw_sub, j << (A.log_s - 1 - s));
            const FeT<Fr> w2a = load_tw(A.tw_sub, j << (A.log_s - 2 - s));
            const FeT<Fr> w2b = load_tw(A.tw_sub, (j + m) << (A.log_s - 2 - s));
#endif
            const Radix4Out o = radix4_lazy(x0, x1, x2, x3, w1, w2a, w2b);
            const FrL y3 = weak(o.y3); // 5 U would reach 7 U in the x2 role of the next pair
#pragma unroll
            for (int l = 0; l < NL; l++) {
                lds[l * E + e0] = o.y0.d[l];
                lds[l * E + e1] = o.y1.d[l];
                lds[l * E + e2] = o.y2.d[l];
                lds[l * E + e3] = y3.d[l];
            }
        }
        // The 64 groups of a wave in stage pair (s, s + 1) are an aligned block of 2^(s + 2) * (64 >> s) = 256 elements as long as s <= 6 (lds_pos keeps
        // such blocks), the SAME block in every such pair: between two of them only the wave's own LDS writes have to be visible to it -- the LDS
        // executes a wave's instructions in order -- and the workgroup's other waves may run ahead or behind (round 5; -DBBGPU_NTT_WG_BARRIERS: A/B).
#ifndef BBGPU_NTT_WG_BARRIERS
        if (s + 2 < last_s && s + 2 <= 6) ntt_wave_sync();
        else
#endif
        __syncthreads();
    }

    // ---- C: last stage pair (even log_s) or last stage (odd) + twist / scaling / canonicalisation + store -----------------------
    if (!odd) {
        const uint32_t m = quarter; // s = log_s - 2
        for (uint32_t gq = tid; gq < ngr; gq += THREADS) {
            uint32_t c, j;
            if (A.store_b_fast) { c = gq & (cols - 1); j = gq >> A.log_cols; } else { j = gq & (quarter - 1); c = gq >> (A.log_s - 2); }
            const uint32_t cb = c * S;
            FrS x0, x1, x2, x3;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                x0.d[l] = lds[l * E + cb + lds_pos(j)];
                x1.d[l] = lds[l * E + cb + lds_pos(j + m)];
                x2.d[l] = lds[l * E + cb + lds_pos(j + 2 * m)];
                x3.d[l] = lds[l * E + cb + lds_pos(j + 3 * m)];
            }
            const uint32_t b = b0 + c;
            if (NTT_DEBUG_SKIP & 1) {
                ntt_finish_store<FLAGS>(A, x0, j, b);
                ntt_finish_store<FLAGS>(A, x1, j + m, b);
                ntt_finish_store<FLAGS>(A, x2, j + 2 * m, b);
                ntt_finish_store<FLAGS>(A, x3, j + 3 * m, b);
            } else {
                // the outputs go straight into the twist / scaling product or into reduce_value(): no renormalisation at all
                const Radix4Out o = radix4_lazy(x0, x1, x2, x3, load_tw(A.tw_sub, j << 1), load_tw(A.tw_sub, j), load_tw(A.tw_sub, j + m));
                ntt_finish_store<FLAGS>(A, o.y0, j, b);
                ntt_finish_store<FLAGS>(A, o.y1, j + m, b);
                ntt_finish_store<FLAGS>(A, o.y2, j + 2 * m, b);
                ntt_finish_store<FLAGS>(A, o.y3, j + 3 * m, b);
            }
        }
    } else {
        const uint32_t half = S >> 1, nbf = cols * half; // s = log_s - 1
        for (uint32_t bf = tid; bf < nbf; bf += THREADS) {
            uint32_t c, j;
            if (A.store_b_fast) { c = bf & (cols - 1); j = bf >> A.log_cols; } else { j = bf & (half - 1); c = bf >> (A.log_s - 1); }
            const uint32_t cb = c * S;
            FrS x, y;
#pragma unroll
            for (int l = 0; l < NL; l++) {
                x.d[l] = lds[l * E + cb + lds_pos(j)];
                y.d[l] = lds[l * E + cb + lds_pos(j + half)];
            }
            const uint32_t b = b0 + c;
            if (NTT_DEBUG_SKIP & 1) {
                ntt_finish_store<FLAGS>(A, x, j, b);
                ntt_finish_store<FLAGS>(A, y, j + half, b);
            } else {
                const auto t = exact_limbs(mul(load_tw(A.tw_sub, j), y));
                ntt_finish_store<FLAGS>(A, assume_bound<5, NTT_VMAX>(add(x, t)), j, b);      // 5 U
                ntt_finish_store<FLAGS>(A, assume_bound<6, NTT_VMAX>(sub(x, t)), j + half, b); // 6 U: still a legal factor
            }
        }
    }
}

// table[k] = base^k * factor   (all Montgomery-261), k < count; canonical, as nine exact 29-bit limbs in a 12-word entry
__global__ void ntt_pow_table_kernel(uint32_t* table, uint32_t count, Limbs9 base, Limbs9 factor)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    Fe<Fr, 1, 2> acc = fe_from<Fr>(factor.d);
    Fe<Fr, 1, 2> b = fe_from<Fr>(base.d);
    for (uint32_t e = k; e; e >>= 1) {
        if (e & 1) acc = mul(acc, b);
        b = sqr(b);
    }
    uint32_t w[8];
    to_canonical(acc, w);
    const Fe<Fr, 1, 6> u = unpack<Fr>(w); // exact limbs of the canonical value
    uint4* q = reinterpret_cast<uint4*>(table + TW_WORDS * (size_t)k);
    q[0] = make_uint4(u.d[0], u.d[1], u.d[2], u.d[3]);
    q[1] = make_uint4(u.d[4], u.d[5], u.d[6], u.d[7]);
    q[2] = make_uint4(u.d[8], 0u, 0u, 0u);
}

// full[k * n2 + b] = w_n^(b k) [* extra[b] or extra[k]], k < n1, b < n2, from the two-level tables; canonical, packed 8 words (the
// layout of pass 1's output).  extra (12-word entries, or null): the column-constant part of a coset scaling (by_k = 0: g^b on the
// way in; by_k = 1: g^-k n^-1 on the way out).
__global__ void ntt_twist_full_kernel(uint32_t* full, const uint32_t* lo, const uint32_t* hi, uint32_t lo_bits, uint32_t log_n2, uint32_t n, const uint32_t* extra,
                                      uint32_t by_k)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t b = g & ((1u << log_n2) - 1), k = g >> log_n2, ex = b * k;
    Fe<Fr, 1, 2> tw = mul(load_tw(lo, ex & ((1u << lo_bits) - 1)), load_tw(hi, ex >> lo_bits));
    if (extra) tw = mul(tw, load_tw(extra, by_k ? k : b));
    uint32_t w[8];
    to_canonical(tw, w);
    store8(full + 8 * (size_t)g, w);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// small host big-int helpers on the 9x29 representation (host-side product code, no oracle involved)
using H = Fe<Fr, 1, 2>;
H h_from(const uint32_t (&a)[NL]) { return fe_from<Fr>(a); }
H h_pow2k(H a, int k) { for (int i = 0; i < k; i++) a = sqr(a); return a; }
Limbs9 to_limbs(const H& a)
{
    // canonicalise so that device tables start from unique representatives
    uint32_t w[8];
    to_canonical(a, w);
    Fe<Fr, 1, 6> u = unpack<Fr>(w);
    Limbs9 r;
    for (int i = 0; i < NL; i++) r.d[i] = u.d[i];
    return r;
}
H h_inv(const H& a)
{
    uint64_t e[4] = { Fr::P64[0] - 2, Fr::P64[1], Fr::P64[2], Fr::P64[3] };
    return pow_u256<Fr>(a, e);
}

struct DomainTables {
    int log2n = -1;
    int log_s1 = 0, log_s2 = 0, lo_bits = 0;
    uint32_t* tw_sub[2][2] = { { nullptr, nullptr }, { nullptr, nullptr } }; // [inverse][pass]
    uint32_t* tw_sub3[2] = { nullptr, nullptr };                             // [inverse]: pass A of a three-pass transform (n > 2^22)
    uint32_t* twist_lo[2] = { nullptr, nullptr };                            // [inverse]
    uint32_t* twist_hi[2] = { nullptr, nullptr };
    uint32_t* twist_full[2] = { nullptr, nullptr };                          // [inverse]: one factor per element (two-pass sizes up to NTT_FULL_TWIST_MAX_LOG2N)
    uint32_t* coset_row[2] = { nullptr, nullptr };        // [0]: (g^n2)^j1, j1 < n1 (pre-scale rows of pass 1); [1]: (g^-n1)^k2, k2 < n2 (post-scale of pass 2; single pass: g^-k n^-1)
    uint32_t* coset_twist[2] = { nullptr, nullptr };      // [0]: w^(j2 k1) g^j2 (coset_fft); [1]: w^-(j2 k1) g^-k1 n^-1 (coset_ifft); two-pass sizes with a full twist table
    uint32_t* scale_lo[2] = { nullptr, nullptr };                            // [0]: g^i, [1]: g^-i * n^-1
    uint32_t* scale_hi[2] = { nullptr, nullptr };
    Limbs9 n_inv;                                                            // Montgomery-261
    size_t bytes = 0;       // device bytes of this table set
    uint64_t last_use = 0;  // LRU clock
};

std::mutex g_mu;
std::unordered_map<int, DomainTables*> g_domains; // keyed by device * 64 + log2n
// The table sets are built on first use of a domain size and kept -- up to a byte budget (BBGPU_NTT_TABLE_BYTES, default 8 GiB: a 2^20 domain holds
// 128 MiB, a 2^22 one 512 MiB).  Beyond it the least recently used sets are dropped and rebuilt when their size comes back (a rebuild is a few
// launches: 2^20 ~0.3 ms, 2^22 ~1 ms).  A set the running call has already fetched (the two domains of a three-pass transform) is never the victim.
size_t g_table_bytes = 0;
uint64_t g_clock = 0;
size_t table_cap() // read when a new table set is built (rare), so a long-lived process can be re-budgeted
{
    const char* e = getenv("BBGPU_NTT_TABLE_BYTES");
    return e ? (size_t)strtoull(e, nullptr, 0) : (size_t)8 << 30;
}
DomainTables* g_building = nullptr; // the set whose allocations are being accounted (build_domain runs under g_mu)
hipError_t table_malloc(void** out, size_t bytes)
{
    hipError_t e = dev_malloc(out, bytes);
    if (e == hipSuccess && g_building) g_building->bytes += bytes;
    return e;
}

hipError_t pow_table(uint32_t** out, uint32_t count, const H& base, const H& factor, hipStream_t st)
{
    hipError_t e = table_malloc((void**)out, (size_t)count * TW_WORDS * 4);
    if (e != hipSuccess) return e;
    ntt_pow_table_kernel<<<(count + 127) / 128, 128, 0, st>>>(*out, count, to_limbs(base), to_limbs(factor));
    return launch_check();
}

// one multiplication per element for the inter-pass twist (a table as large as the vector) instead of two (two sqrt(n)-sized tables); BBGPU_NTT_FULL_TWIST=0: A/B
bool full_twist_enabled()
{
    static const bool on = [] { const char* e = getenv("BBGPU_NTT_FULL_TWIST"); return !e || atoi(e) != 0; }();
    return on;
}

hipError_t build_domain(DomainTables* D, int log2n, hipStream_t st)
{
    D->log2n = log2n;
    // split: sub-transform sizes as balanced as possible, both <= 2^NTT_MAX_LOG_SUB
    if (log2n <= NTT_MAX_LOG_SUB) {
        D->log_s1 = log2n;
        D->log_s2 = 0;
    } else {
        D->log_s2 = log2n / 2;
        D->log_s1 = log2n - D->log_s2;
    }
    const bool three = log2n > 2 * NTT_MAX_LOG_SUB;
    if (three) D->log_s1 = D->log_s2 = 0; // the row transforms use the tables of their own (size-m) domain
    D->lo_bits = (log2n + 1) / 2;
    const H one = fe_one<Fr>();
    H root = h_pow2k(h_from(Fr::ROOT28), 28 - log2n);          // w_n   (field.hpp:487-494 semantics)
    H root_inv = h_pow2k(h_from(Fr::ROOT28_INV), 28 - log2n);  // w_n^-1
    // n^-1 = (2^-1)^log2n
    H ninv = one;
    {
        H half = h_inv(weak(add(fe_one<Fr>(), fe_one<Fr>())));
        for (int i = 0; i < log2n; i++) ninv = mul(ninv, half);
    }
    D->n_inv = to_limbs(ninv);
    hipError_t e;
    for (int inv = 0; inv < 2; inv++) {
        const H w = inv ? root_inv : root;
        // sub-transform twiddles: w_S = w_n^(n/S)
        const int logs[2] = { D->log_s1, D->log_s2 };
        for (int p = 0; p < 2; p++) {
            if (logs[p] < 1) continue;
            H ws = h_pow2k(w, log2n - logs[p]);
            uint32_t cnt = logs[p] >= 1 ? (1u << (logs[p] - 1)) : 1;
            if ((e = pow_table(&D->tw_sub[inv][p], cnt, ws, one, st)) != hipSuccess) return e;
        }
        if (three) { // w_n1 = w_n^(n / n1), n1 = 2^(log2n - 2 (log2n / 3))
            const int l1 = log2n - 2 * (log2n / 3);
            if ((e = pow_table(&D->tw_sub3[inv], 1u << (l1 - 1), h_pow2k(w, log2n - l1), one, st)) != hipSuccess) return e;
        }
        if ((e = pow_table(&D->twist_lo[inv], 1u << D->lo_bits, w, one, st)) != hipSuccess) return e;
        if ((e = pow_table(&D->twist_hi[inv], 1u << (log2n - D->lo_bits), h_pow2k(w, D->lo_bits), one, st)) != hipSuccess) return e;
        if (full_twist_enabled() && !three && D->log_s2 > 0 && log2n <= NTT_FULL_TWIST_MAX_LOG2N) {
            const uint32_t n = 1u << log2n;
            if ((e = table_malloc((void**)&D->twist_full[inv], (size_t)n * 32)) != hipSuccess) return e;
            ntt_twist_full_kernel<<<(n + 255) / 256, 256, 0, st>>>(D->twist_full[inv], D->twist_lo[inv], D->twist_hi[inv], (uint32_t)D->lo_bits, (uint32_t)D->log_s2, n, nullptr, 0u);
            if ((e = launch_check()) != hipSuccess) return e;
        }
    }
    // coset scale tables: [0] g^i ; [1] g^-i * n^-1   (generator 5: fr.hpp:66-74)
    H g = h_from(Fr::GEN5), gi = h_from(Fr::GEN5_INV);
    if ((e = pow_table(&D->scale_lo[0], 1u << D->lo_bits, g, one, st)) != hipSuccess) return e;
    if ((e = pow_table(&D->scale_hi[0], 1u << (log2n - D->lo_bits), h_pow2k(g, D->lo_bits), one, st)) != hipSuccess) return e;
    if ((e = pow_table(&D->scale_lo[1], 1u << D->lo_bits, gi, one, st)) != hipSuccess) return e;
    if ((e = pow_table(&D->scale_hi[1], 1u << (log2n - D->lo_bits), h_pow2k(gi, D->lo_bits), ninv, st)) != hipSuccess) return e;
    // coset scalings at one multiplication per element (fused kernel, FLAGS 64 / 128)
    if (!three && D->log_s1 >= 4) {
        if (D->log_s2 == 0) { // single pass: plain row tables over the whole index
            if ((e = pow_table(&D->coset_row[0], 1u << log2n, g, one, st)) != hipSuccess) return e;
            if ((e = pow_table(&D->coset_row[1], 1u << log2n, gi, ninv, st)) != hipSuccess) return e;
        } else if (D->twist_full[0]) {
            const uint32_t n = 1u << log2n;
            if ((e = pow_table(&D->coset_row[0], 1u << D->log_s1, h_pow2k(g, D->log_s2), one, st)) != hipSuccess) return e;   // (g^n2)^j1
            if ((e = pow_table(&D->coset_row[1], 1u << D->log_s2, h_pow2k(gi, D->log_s1), one, st)) != hipSuccess) return e;  // (g^-n1)^k2
            uint32_t *gb = nullptr, *gk = nullptr; // g^b, b < n2 and g^-k n^-1, k < n1: folded into the coset twist tables, then dropped
            const size_t before_tmp = D->bytes;
            // (a table whose launch check failed is allocated all the same: both temporaries are freed on every way out -- found by the injected launch failures of round 5)
            if ((e = pow_table(&gb, 1u << D->log_s2, g, one, st)) != hipSuccess) { (void)dev_free(gb); return e; }
            if ((e = pow_table(&gk, 1u << D->log_s1, gi, ninv, st)) != hipSuccess) { (void)dev_free(gb); (void)dev_free(gk); return e; }
            const size_t tmp_bytes = D->bytes - before_tmp;
            for (int inv = 0; inv < 2 && e == hipSuccess; inv++) {
                if ((e = table_malloc((void**)&D->coset_twist[inv], (size_t)n * 32)) != hipSuccess) break;
                ntt_twist_full_kernel<<<(n + 255) / 256, 256, 0, st>>>(D->coset_twist[inv], D->twist_lo[inv], D->twist_hi[inv], (uint32_t)D->lo_bits, (uint32_t)D->log_s2, n,
                                                                       inv ? gk : gb, (uint32_t)inv);
                e = launch_check();
            }
            (void)hipStreamSynchronize(st);
            (void)dev_free(gb);
            (void)dev_free(gk);
            D->bytes -= tmp_bytes;
            if (e != hipSuccess) return e;
        }
    }
    // the tables are published to every later caller, whatever stream it runs on: they must be COMPLETE before get_domain returns
    // (once per domain size; an event per table set would only save this one wait)
    return hipStreamSynchronize(st);
}

void free_domain(DomainTables* D)
{
    for (int i = 0; i < 2; i++) {
        for (int p = 0; p < 2; p++) if (D->tw_sub[i][p]) (void)dev_free(D->tw_sub[i][p]);
        if (D->tw_sub3[i]) (void)dev_free(D->tw_sub3[i]);
        if (D->twist_lo[i]) (void)dev_free(D->twist_lo[i]);
        if (D->twist_hi[i]) (void)dev_free(D->twist_hi[i]);
        if (D->twist_full[i]) (void)dev_free(D->twist_full[i]);
        if (D->coset_row[i]) (void)dev_free(D->coset_row[i]);
        if (D->coset_twist[i]) (void)dev_free(D->coset_twist[i]);
        if (D->scale_lo[i]) (void)dev_free(D->scale_lo[i]);
        if (D->scale_hi[i]) (void)dev_free(D->scale_hi[i]);
    }
    delete D;
}

// keep: a table set the running call fetched earlier (never evicted by this fetch), or null
hipError_t get_domain(int log2n, hipStream_t st, DomainTables** out, const DomainTables* keep = nullptr)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_domains.find(dev * 64 + log2n);
    if (it != g_domains.end()) {
        it->second->last_use = ++g_clock;
        *out = it->second;
        return hipSuccess;
    }
    DomainTables* D = new DomainTables();
    g_building = D;
    hipError_t e = build_domain(D, log2n, st);
    g_building = nullptr;
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st); // kernels of the tables built so far
        free_domain(D);                 // including the partially built set
        return e;
    }
    D->last_use = ++g_clock;
    g_table_bytes += D->bytes;
    g_domains[dev * 64 + log2n] = D;
    // over the budget: drop the least recently used sets (not this one, not `keep`).  Transforms that still use a victim may be in flight on any
    // stream: drain the device first (rare path: a process that walks through more domain sizes than the budget holds)
    bool drained = false;
    while (g_table_bytes > table_cap()) {
        auto victim = g_domains.end();
        for (auto jt = g_domains.begin(); jt != g_domains.end(); ++jt)
            if (jt->second != D && jt->second != keep && (victim == g_domains.end() || jt->second->last_use < victim->second->last_use)) victim = jt;
        if (victim == g_domains.end()) break; // the sets of the running call alone exceed the budget: they stay
        if (!drained) (void)hipDeviceSynchronize();
        drained = true;
        g_table_bytes -= victim->second->bytes;
        free_domain(victim->second);
        g_domains.erase(victim);
    }
    *out = D;
    return hipSuccess;
}


bool fused_enabled()
{
    static const bool on = [] { const char* e = getenv("BBGPU_NTT_FUSED"); return !e || atoi(e) != 0; }();
    return on;
}

// elements per workgroup tile (the fused kernel): 1024 (36 KiB, four workgroups of 256 threads per CU), 2048 (72 KiB, two of 512) or 4096
// (144 KiB, one of 1024: the strided side of a pass then moves rows twice as wide).  Measured (tools/ntt_sizes.py, one box, fft):
//   tile   2^12    2^14    2^16    2^18    2^20    2^22    2^24
//   1024  0.031   0.033   0.035   0.050   0.133   0.531   2.17  ms   <- below 2^20 transforms are latency-bound: twice the workgroups, half the barrier width
//   2048  0.055   0.052   0.053   0.064   0.130   0.527   2.16
//   4096  0.089   0.100   0.092   0.102   0.131   0.543   2.38       <- wider rows buy nothing: the passes are not bound by coalescing
// BBGPU_NTT_TILE overrides.
uint32_t ntt_tile_elems(int log2n)
{
    static const int forced = [] { const char* e = getenv("BBGPU_NTT_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 1024 || forced == 2048 || forced == 4096) return (uint32_t)forced;
    return log2n > 0 && log2n < 20 ? NTT_LDS_ELEMS / 2 : NTT_LDS_ELEMS;
}

template <int FLAGS> hipError_t launch_pass(const NttPassArgs& A, hipStream_t st)
{
    const uint32_t S = 1u << A.log_s, blocks = (1u << A.log_b) / A.cols;
    const size_t lds = (size_t)A.cols * S * NL * 4; // the unfused kernel; the fused instantiations take their whole tile (constant plane stride)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ntt_pass_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_LDS_ELEMS * NL * 4);
        (void)hipFuncSetAttribute((const void*)ntt_pass_fused_kernel<FLAGS, NTT_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_LDS_ELEMS * NL * 4 + NTT_TW_LDS_ENTRIES * TW_WORDS * 4);
        (void)hipFuncSetAttribute((const void*)ntt_pass_fused_kernel<FLAGS, 2 * NTT_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NTT_LDS_ELEMS * NL * 4);
        (void)hipFuncSetAttribute((const void*)ntt_pass_fused_kernel<FLAGS, NTT_THREADS / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_LDS_ELEMS * NL * 4);
        attr_set = true;
    }
    // sub-transforms of 16 points and more take the kernel with load / store fused into the first / last stage pair (BBGPU_NTT_FUSED=0: A/B)
    if (fused_enabled() && A.log_s >= 4) {
        NttPassArgs B = A;
        B.store_b_fast = (A.out_sb == 1 && A.cols > 1) ? 1u : 0u;
        if ((size_t)A.cols * S > (size_t)NTT_LDS_ELEMS) // double tile: one workgroup of 1024 threads per CU (144 KiB of LDS), rows twice as wide
            ntt_pass_fused_kernel<FLAGS, 2 * NTT_THREADS><<<dim3(blocks, A.batch ? A.batch : 1), 2 * NTT_THREADS, (size_t)8 * NTT_THREADS * NL * 4, st>>>(B);
        else if (A.half_tile && (size_t)A.cols * S <= (size_t)NTT_LDS_ELEMS / 2) { // half tile: four workgroups of 256 threads per CU
            // BBGPU_NTT_LDS_PAD (tuning experiment): bytes of LDS requested on top of the tile, e.g. 36864 keeps TWO half-tile workgroups per CU (two waves per SIMD)
            static const size_t pad = [] { const char* e = getenv("BBGPU_NTT_LDS_PAD"); return e ? std::min<size_t>((size_t)2 * NTT_THREADS * NL * 4, (size_t)strtoull(e, nullptr, 0)) : (size_t)0; }(); // tuning experiment: extra LDS bytes per half-tile workgroup (36864 = two workgroups per CU)
            ntt_pass_fused_kernel<FLAGS, NTT_THREADS / 2><<<dim3(blocks, A.batch ? A.batch : 1), NTT_THREADS / 2, (size_t)2 * NTT_THREADS * NL * 4 + pad, st>>>(B);
        }
        else {
#ifndef BBGPU_NTT_TW_GLOBAL
            // Round 5, one box, steady state (profiles/r05_ntt_twlds_ab.txt): 2^22 fft 0.416 -> 0.396 ms (-3.5 ... 5.7 % by kind: its 1024-entry tables, 48 KiB, do not fit the 32 KiB L1),
            // 2^20 level (512 entries do); the 256-thread instance (sizes below 2^20) has no room for the table beside four 36 KiB tiles per CU and keeps the global loads
            B.tw_lds = A.log_s >= 8 ? 1u : 0u; // 128 staged entries cover the pairs up to s = 6 of sub-transforms of 256 points and more
            ntt_pass_fused_kernel<FLAGS, NTT_THREADS><<<dim3(blocks, A.batch ? A.batch : 1), NTT_THREADS, (size_t)4 * NTT_THREADS * NL * 4 + (B.tw_lds ? NTT_TW_LDS_ENTRIES * TW_WORDS * 4 : 0), st>>>(B);
#else
            ntt_pass_fused_kernel<FLAGS, NTT_THREADS><<<dim3(blocks, A.batch ? A.batch : 1), NTT_THREADS, (size_t)4 * NTT_THREADS * NL * 4, st>>>(B);
#endif
        }
    } else {
        if ((size_t)A.cols * S > (size_t)NTT_LDS_ELEMS) return hipErrorInvalidValue; // the double tile exists in the fused kernel only
        ntt_pass_kernel<FLAGS><<<dim3(blocks, A.batch ? A.batch : 1), NTT_THREADS, lds, st>>>(A);
    }
    return launch_check();
}

hipError_t dispatch(int flags, const NttPassArgs& A, hipStream_t st)
{
    switch (flags) {
#define CASE(F) case F: return launch_pass<F>(A, st);
        CASE(2) CASE(3) CASE(34) CASE(35) CASE(98)        // pass 1: twist (two-level tables / full table), optionally pre-scaled (two-level / row table)
        CASE(16) CASE(17) CASE(20) CASE(24) CASE(25) CASE(28) // last pass variants
        CASE(80) CASE(88) CASE(144)                       // single pass with a row-table pre-scale (16|64, 24|64); post-scale by a row table (16|128)
#undef CASE
    default: return hipErrorInvalidValue;
    }
}

} // namespace

// n = 2^23 .. 2^28 (the field's two-adicity, fr.hpp:60-63): n = n1 * m, three passes over HBM
//   A: for every column j' < m: n1-point transform over j1 (stride m) IN PLACE (a workgroup owns whole columns), twist by w_n^(k1 j')
//   B, C: for every row k1 < n1: the m-point transform of the contiguous row (the two-pass scheme above with the tables of the size-m
//         domain, w_m = w_n^n1), rows side by side in one launch (blockIdx.y = k1); results land transposed: X[k1 + n1 k']
// Coset pre-scaling rides on pass A's load, post-scaling / constants on pass C's store (natural index k1 + n1 k').
static int ntt_device_three_pass(uint64_t* d_coeffs, uint64_t* d_scratch, int log2n, int kind, const uint64_t* constant_m256, hipStream_t st)
{
    const int lm = 2 * (log2n / 3), l1 = log2n - lm; // row transforms split evenly; every sub-transform <= 2^10
    DomainTables *D, *Dm;
    if (get_domain(log2n, st, &D) != hipSuccess || get_domain(lm, st, &Dm, D) != hipSuccess) return BBGPU_ERR_HIP;
    const bool inverse = (kind == BBGPU_IFFT || kind == BBGPU_COSET_IFFT || kind == BBGPU_IFFT_WITH_CONSTANT);
    const bool pre = (kind == BBGPU_COSET_FFT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    const bool post_table = (kind == BBGPU_COSET_IFFT);
    const bool has_const = (kind == BBGPU_FFT_WITH_CONSTANT || kind == BBGPU_IFFT_WITH_CONSTANT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    if (has_const && !constant_m256) return BBGPU_ERR_ARG;
    bool post_const = false;
    H pc = fe_one<Fr>();
    if (has_const) {
        uint32_t w[8];
        for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)constant_m256[i]; w[2 * i + 1] = (uint32_t)(constant_m256[i] >> 32); }
        pc = m256_to_m261<Fr>(unpack<Fr>(w));
        post_const = true;
    }
    if (kind == BBGPU_IFFT || kind == BBGPU_IFFT_WITH_CONSTANT) {
        pc = mul(pc, fe_from<Fr>(D->n_inv.d));
        post_const = true;
    }
    const Limbs9 pcl = to_limbs(pc);
    const uint32_t n1 = 1u << l1, m = 1u << lm, m1 = 1u << Dm->log_s1, m2 = 1u << Dm->log_s2;
    NttPassArgs A{};
    A.half_tile = ntt_tile_elems(log2n) == NTT_LDS_ELEMS / 2 ? 1u : 0u;
    A.xcd_remap = 1;
    for (int i = 0; i < NL; i++) A.post_const[i] = pcl.d[i];
    hipError_t e;
    // pass A: columns of the n1 x m matrix, in place
    A.batch = 1;
    A.in = (const uint32_t*)d_coeffs;
    A.out = (uint32_t*)d_coeffs;
    A.tw_sub = D->tw_sub3[inverse] ;
    A.lo_bits = D->lo_bits;
    A.twist_lo = D->twist_lo[inverse]; A.twist_hi = D->twist_hi[inverse];
    A.scale_lo = D->scale_lo[0]; A.scale_hi = D->scale_hi[0];
    A.log_s = (uint32_t)l1; A.log_b = (uint32_t)lm;
    A.cols = std::max<uint32_t>(1u, ntt_tile_elems(log2n) >> l1); if (A.cols > m) A.cols = m;
    A.log_cols = 31 - __builtin_clz(A.cols);
    A.in_sa = m; A.in_sb = 1; A.out_sa = m; A.out_sb = 1; A.b_fast = 1;
    if ((e = dispatch(2 | (pre ? 1 : 0), A, st)) != hipSuccess) return BBGPU_ERR_HIP;
    // pass B: columns of every row's m1 x m2 matrix, coeffs -> scratch (same layout), twist by w_m^(k j)
    A.batch = n1;
    A.in = (const uint32_t*)d_coeffs;
    A.out = (uint32_t*)d_scratch;
    A.in_bstride = A.out_bstride = (size_t)m * 8;
    A.tw_sub = Dm->tw_sub[inverse][0];
    A.lo_bits = Dm->lo_bits;
    A.twist_lo = Dm->twist_lo[inverse]; A.twist_hi = Dm->twist_hi[inverse];
    A.log_s = Dm->log_s1; A.log_b = Dm->log_s2;
    A.cols = std::max<uint32_t>(1u, ntt_tile_elems(log2n) >> Dm->log_s1); if (A.cols > m2) A.cols = m2;
    A.log_cols = 31 - __builtin_clz(A.cols);
    A.in_sa = m2; A.in_sb = 1; A.out_sa = m2; A.out_sb = 1; A.b_fast = 1;
    if ((e = dispatch(2, A, st)) != hipSuccess) return BBGPU_ERR_HIP;
    // pass C: rows of every row's matrix, scratch -> coeffs; element k' = k1m + m1 k2m of row k1 goes to X[k1 + n1 k']
    A.in = (const uint32_t*)d_scratch;
    A.out = (uint32_t*)d_coeffs;
    A.in_bstride = (size_t)m * 8;
    A.out_bstride = 8;   // row k1 starts one element further
    A.nat_bstep = 1;
    A.tw_sub = Dm->tw_sub[inverse][1];
    A.lo_bits = D->lo_bits; // the post-scale tables are those of the size-n domain
    A.scale_lo = D->scale_lo[post_table ? 1 : 0]; A.scale_hi = D->scale_hi[post_table ? 1 : 0];
    A.log_s = Dm->log_s2; A.log_b = Dm->log_s1;
    A.cols = std::max<uint32_t>(1u, ntt_tile_elems(log2n) >> Dm->log_s2); if (A.cols > m1) A.cols = m1;
    A.log_cols = 31 - __builtin_clz(A.cols);
    A.in_sa = 1; A.in_sb = m2; A.out_sa = m1 * n1; A.out_sb = n1; A.b_fast = 0;
    const int last_flags = 16 | (post_table ? 4 : 0) | (post_const ? 8 : 0);
    if ((e = dispatch(last_flags, A, st)) != hipSuccess) return BBGPU_ERR_HIP;
    return BBGPU_OK;
}

// kind: bbgpu_ntt_kind; d_coeffs: n x 32 B device buffer, transformed in place; d_scratch: n x 32 B (only n > 2^11)
int ntt_device(uint64_t* d_coeffs, uint64_t* d_scratch, int log2n, int kind, const uint64_t* constant_m256, hipStream_t st)
{
    return ntt_device_batch(d_coeffs, (size_t)1 << log2n, 1, d_scratch, log2n, kind, constant_m256, st);
}
// `batch` independent transforms of the same size and kind in the same launches (small transforms are latency-bound: a
// 2^18-point pass occupies half the CUs for ~40 us, three of them side by side take no longer).  Transform j lives at
// d_coeffs + j * stride_elems elements; d_scratch: batch * n x 32 B.
int ntt_device_batch(uint64_t* d_coeffs, size_t stride_elems, int batch, uint64_t* d_scratch, int log2n, int kind, const uint64_t* constant_m256,
                     hipStream_t st)
{
    if (batch < 1) return BBGPU_ERR_ARG;
    if (log2n < 1 || log2n > NTT_MAX_LOG2N) return BBGPU_ERR_SIZE;
    if (log2n > 2 * NTT_MAX_LOG_SUB) { // three HBM passes, one transform at a time
        if (!d_scratch) return BBGPU_ERR_ARG;
        for (int j = 0; j < batch; j++) {
            const int rc = ntt_device_three_pass(d_coeffs + (size_t)j * stride_elems * 4, d_scratch, log2n, kind, constant_m256, st);
            if (rc != BBGPU_OK) return rc;
        }
        return BBGPU_OK;
    }
    DomainTables* D;
    if (get_domain(log2n, st, &D) != hipSuccess) return BBGPU_ERR_HIP;
    const bool inverse = (kind == BBGPU_IFFT || kind == BBGPU_COSET_IFFT || kind == BBGPU_IFFT_WITH_CONSTANT);
    const bool pre = (kind == BBGPU_COSET_FFT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    const bool post_table = (kind == BBGPU_COSET_IFFT);
    const bool has_const = (kind == BBGPU_FFT_WITH_CONSTANT || kind == BBGPU_IFFT_WITH_CONSTANT || kind == BBGPU_COSET_FFT_WITH_CONSTANT);
    if (has_const && !constant_m256) return BBGPU_ERR_ARG;

    // constant applied at the last store (Montgomery-261): c (converted from the caller's 2^256 form), times n^-1 for
    // ifft; coset_ifft gets its n^-1 from the scale_hi table
    bool post_const = false;
    H pc = fe_one<Fr>();
    if (has_const) {
        uint32_t w[8];
        for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)constant_m256[i]; w[2 * i + 1] = (uint32_t)(constant_m256[i] >> 32); }
        pc = m256_to_m261<Fr>(unpack<Fr>(w));
        post_const = true;
    }
    if (kind == BBGPU_IFFT || kind == BBGPU_IFFT_WITH_CONSTANT) {
        pc = mul(pc, fe_from<Fr>(D->n_inv.d));
        post_const = true;
    }
    const Limbs9 pcl = to_limbs(pc);

    const uint32_t n1 = 1u << D->log_s1, n2 = 1u << D->log_s2;
    NttPassArgs A{};
    A.half_tile = ntt_tile_elems(log2n) == NTT_LDS_ELEMS / 2 ? 1u : 0u;
    A.batch = (uint32_t)batch;
    {
        // measured (tools/ntt_sizes.py, fft): 2^22 0.617 -> 0.584 ms, 2^21 0.320 -> 0.302, 2^20 0.153 -> 0.149, 2^18 0.0695 -> 0.0707 (slightly worse)
        static const int remap = [] { const char* e = getenv("BBGPU_NTT_XCD"); return e ? atoi(e) : -1; }(); // tuning knob: 0 / 1 force
        A.xcd_remap = remap >= 0 ? (uint32_t)remap : (log2n >= 20 ? 1u : 0u);
    }
    A.lo_bits = D->lo_bits;
    A.twist_lo = D->twist_lo[inverse];
    A.twist_hi = D->twist_hi[inverse];
    A.scale_lo = D->scale_lo[post_table ? 1 : 0];
    A.scale_hi = D->scale_hi[post_table ? 1 : 0];
    for (int i = 0; i < NL; i++) A.post_const[i] = pcl.d[i];
    const int last_flags = 16 | (post_table ? 4 : 0) | (post_const ? 8 : 0);
    hipError_t e;
    if (D->log_s2 == 0) { // single pass, everything in one workgroup's LDS
        A.in = (const uint32_t*)d_coeffs;
        A.out = (uint32_t*)d_coeffs;
        A.in_bstride = A.out_bstride = stride_elems * 8;
        A.tw_sub = D->tw_sub[inverse][0];
        A.log_s = D->log_s1; A.log_b = 0; A.cols = 1; A.log_cols = 0;
        A.in_sa = 1; A.in_sb = 0; A.out_sa = 1; A.out_sb = 0; A.b_fast = 0;
        int flags = last_flags | (pre ? 1 : 0);
        if (fused_enabled() && D->coset_row[0] && (pre || post_table)) { // coset scalings from a per-index table: one multiplication
            A.row_scale = D->coset_row[post_table ? 1 : 0];
            flags = pre ? ((last_flags & ~4) | 64) : (16 | 128);
        }
        e = dispatch(flags, A, st);
        return e == hipSuccess ? BBGPU_OK : BBGPU_ERR_HIP;
    }
    if (!d_scratch) return BBGPU_ERR_ARG;
    // pass 1: columns (a = j1, b = j2), coeffs -> scratch, same layout
    A.in = (const uint32_t*)d_coeffs;
    A.out = (uint32_t*)d_scratch;
    A.in_bstride = stride_elems * 8;
    A.out_bstride = ((size_t)1 << log2n) * 8;
    A.tw_sub = D->tw_sub[inverse][0];
    A.log_s = D->log_s1; A.log_b = D->log_s2;
    A.cols = std::max<uint32_t>(1u, ntt_tile_elems(log2n) >> D->log_s1); if (A.cols > n2) A.cols = n2;
    A.log_cols = 31 - __builtin_clz(A.cols);
    A.in_sa = n2; A.in_sb = 1; A.out_sa = n2; A.out_sb = 1; A.b_fast = 1;
    A.twist_full = D->twist_full[inverse];
    int flags1 = 2 | (pre ? 1 : 0) | (A.twist_full ? 32 : 0), flags2 = last_flags;
    if (fused_enabled() && D->coset_twist[0] && (pre || post_table)) {
        // coset variants at one extra multiplication per element: the column-constant half of the scaling rides in the coset twist table,
        // the other half is a row table of pass 1 (coset_fft) or of pass 2's outputs (coset_ifft)
        A.twist_full = D->coset_twist[post_table ? 1 : 0];
        if (pre) { A.row_scale = D->coset_row[0]; flags1 = 2 | 32 | 64; }
        else { flags1 = 2 | 32; flags2 = 16 | 128; }
    }
    if ((e = dispatch(flags1, A, st)) != hipSuccess) return BBGPU_ERR_HIP;
    // pass 2: rows (a = j2, b = k1), scratch -> coeffs transposed: X[k1 + n1 * k2]
    A.in = (const uint32_t*)d_scratch;
    A.out = (uint32_t*)d_coeffs;
    A.in_bstride = ((size_t)1 << log2n) * 8;
    A.out_bstride = stride_elems * 8;
    A.tw_sub = D->tw_sub[inverse][1];
    A.log_s = D->log_s2; A.log_b = D->log_s1;
    A.cols = std::max<uint32_t>(1u, ntt_tile_elems(log2n) >> D->log_s2); if (A.cols > n1) A.cols = n1;
    A.log_cols = 31 - __builtin_clz(A.cols);
    A.in_sa = 1; A.in_sb = n2; A.out_sa = n1; A.out_sb = 1; A.b_fast = 0;
    if (flags2 & 128) A.row_scale = D->coset_row[1];
    if ((e = dispatch(flags2, A, st)) != hipSuccess) return BBGPU_ERR_HIP;
    return BBGPU_OK;
}

size_t ntt_table_bytes(size_t* cap_out, int* sets_out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (cap_out) *cap_out = table_cap();
    if (sets_out) *sets_out = (int)g_domains.size();
    return g_table_bytes;
}

void ntt_release_tables()
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_domains) free_domain(kv.second);
    g_domains.clear();
    g_table_bytes = 0;
}

} // namespace bbgpu
