// poly.hip -- device-resident polynomial helpers over Fr for gfx950: the scan-shaped and pointwise O(n) pieces of the
// PLONK prover that sit between the NTTs and the MSMs.
//
// Replaces, on resident vectors, the reference's serial / OpenMP-chunked CPU loops
//   polynomial_arithmetic::evaluate                              polynomials/polynomial_arithmetic.cpp:337-373
//   polynomial_arithmetic::compute_kate_opening_coefficients     :562-591
//   polynomial_arithmetic::compute_lagrange_polynomial_fft       :381-476
//   polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial :478-560
//   fr::batch_invert                                             fields/field.hpp:503-522
//   the grand-product prefix products                            waffle/proof_system/prover/prover.cpp:194-202
//   compute_permutation_lagrange_base_single                     waffle/proof_system/permutation.hpp:15-87
// and provides the fused pointwise kernels of the prover rounds (prover.cpp:135-222,224-300,302-403,461-463,520-528,567-595,
// arithmetic_widget.cpp:66-104,106-126).
//
// Data format: every vector is the reference's own (n x 4 x u64, Montgomery 2^256), read as any representative below 2^256
// and always written canonical.  Kernel constants arrive as canonical 9 x 29-bit limbs in one of two forms (host_fr.hpp):
//   *_m256 : x * 2^256, the memory form -- adds to / subtracts from loaded values;
//   *_m261 : x * 2^261 -- a multiplier: mont261(a * 2^256, x * 2^261) = a x * 2^256 stays in memory form.
// A raw product of k memory-form values carries 2^(261 - 5k); one multiplication by FIXk = 2^(256 + 5k) repairs it.
// Sequences like g * w^i are generated in-kernel in the 2^261 form: thread t raises the base to t from a table of
// base^(2^j) and then strides by base^T, so no root table is read from memory.
//
// Scans: the serial chains of the reference (prefix products, Horner suffix sums) become three-phase workgroup scans:
// per-thread runs of 8 elements, a Hillis-Steele doubling over the 256 thread partials in LDS, a single-workgroup scan
// of the block partials, and a final pass that applies the carries.  Exact field arithmetic makes the result independent
// of the association order, so outputs are bit-identical to the serial loops.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "bbgpu_internal.h"
#include "fe.hpp"
#include "host_fr.hpp"
#include "poly.h"

namespace bbgpu {
namespace poly {

using Fr = FrP;
using FrC = FeT<Fr>;      // canonical constant
using FrV = Fe<Fr, 1, 6>; // a loaded memory value (< 2^256 < 6r)
using FrM = Fe<Fr, 1, 2>; // a fresh product

constexpr int PT = 256;  // threads per workgroup of the pointwise kernels
constexpr int RUN = 8;   // elements per thread in the scan kernels
constexpr int SCAN_T = 256;
constexpr int SCAN_BLOCK = SCAN_T * RUN; // 2048 elements per workgroup

#define HIPCHK(x)                                                                                                      \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));                                \
            return BBGPU_ERR_HIP;                                                                                      \
        }                                                                                                              \
    } while (0)

// ---- device helpers -------------------------------------------------------------------------------------------------
__device__ __forceinline__ FrV ldv(const uint32_t* __restrict__ p, size_t i)
{
    const uint4* q = reinterpret_cast<const uint4*>(p + i * 8);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    return unpack<Fr>(w);
}
template <int L, int V> __device__ __forceinline__ void stv(uint32_t* __restrict__ p, size_t i, const Fe<Fr, L, V>& v)
{
    uint32_t w[8];
    to_canonical(v, w);
    uint4* q = reinterpret_cast<uint4*>(p + i * 8);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ FrC cst(const Limbs9& c) { return fe_from<Fr>(c.d); }
__device__ __forceinline__ FrC fix2() { return fe_from<Fr>(Fr::M256_TO_M261); }
__device__ __forceinline__ FrC fix3() { return fe_from<Fr>(Fr::FIX3); }
__device__ __forceinline__ FrC fix4() { return fe_from<Fr>(Fr::FIX4); }
// value * value -> memory form
template <int L1, int V1, int L2, int V2> __device__ __forceinline__ FrM mulv(const Fe<Fr, L1, V1>& a, const Fe<Fr, L2, V2>& b)
{
    return mul(mul(a, b), fix2());
}
template <class F, int L, int V> __device__ __forceinline__ Fe<F, 1, 2> tight2(const Fe<F, L, V>& a)
{
    // squeeze a lazy value back to the (1, 2) loop-carried type with one multiplication by one (2^261 form)
    return mul(a, fe_from<F>(F::ONE));
}
using FrH = Fe<Fr, 2, 8>; // loop-carried Horner accumulator: (product) + (loaded value), fed straight into the next multiply

// start * base^e, start in either form (the result keeps it); T.p[j] = base^(2^j) in the 2^261 form
__device__ __forceinline__ FrM pow_tab(const PowTab& T, uint32_t e, const Limbs9& start)
{
    FrM acc = mul(cst(start), fe_from<Fr>(Fr::ONE));
    for (int j = 0; e; j++, e >>= 1)
        if (e & 1) acc = mul(acc, cst(T.p[j]));
    return acc;
}

// ---- small utility kernels ------------------------------------------------------------------------------------------
// out[i] = start * base^i  (start_m256 -> a table in memory form)
__global__ void __launch_bounds__(PT) k_powers(uint32_t* __restrict__ out, uint32_t n, PowTab T, Limbs9 start_m256, Limbs9 step_m261)
{
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    FrM x = pow_tab(T, t, start_m256);
    const FrC step = cst(step_m261);
    for (uint32_t i = t; i < n; i += nt) {
        stv(out, i, x);
        x = mul(x, step);
    }
}
// dst[0..n_dst) = src[0..n_src) followed by zeros (polynomial(other, size): polynomial.cpp:48-66), canonicalising
__global__ void __launch_bounds__(PT) k_copy_pad(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n_src, uint32_t n_dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst) return;
    uint4* q = reinterpret_cast<uint4*>(dst + (size_t)i * 8);
    if (i < n_src) {
        const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)i * 8);
        q[0] = s[0];
        q[1] = s[1];
    } else {
        q[0] = make_uint4(0, 0, 0, 0);
        q[1] = make_uint4(0, 0, 0, 0);
    }
}
// a[i] += b[i]
__global__ void __launch_bounds__(PT) k_add_inplace(uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stv(a, i, add(ldv(a, i), ldv(b, i)));
}
// out[i] = a[i] * b[i] * c   (c in the 2^261 form, already carrying the FIX2 factor: c' = c * 2^5)
__global__ void __launch_bounds__(PT) k_mul2c(uint32_t* __restrict__ out, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n,
                                            Limbs9 c_fix_m261)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stv(out, i, mul(mul(ldv(a, i), ldv(b, i)), cst(c_fix_m261)));
}
// out[i] = a[i] * b[i]     (polynomial_arithmetic::mul, :328-335)
__global__ void __launch_bounds__(PT) k_mul(uint32_t* __restrict__ out, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stv(out, i, mulv(ldv(a, i), ldv(b, i)));
}

// ---- scans ----------------------------------------------------------------------------------------------------------
// MODE 0: running product  y_i = prod of x over the scanned prefix        (identity 1, combine a * b)
// MODE 1: Horner sums      y_i = sum_{j >= i} x_j z^(j - i)               (suffix; combine y_i = x_i + z^len(i..) * y_next)
// Both are scans of an associative operation; DIR selects prefix (0) or suffix (1) for MODE 0, MODE 1 is always suffix.
// Values travel between phases in memory form (canonical), so the three phases compose exactly.
struct ScanArgs {
    const uint32_t* in;   // n elements
    uint32_t* out;        // n elements (phase 3)
    uint32_t* tpart;      // per-thread partial, exclusive within its block: ceil(n / RUN) elements
    uint32_t* bpart;      // per-block total: ceil(n / SCAN_BLOCK) elements
    const uint32_t* bcarry; // per-block exclusive carry (phase 3)
    uint32_t n;
    uint32_t reverse;     // MODE 0: 1 = suffix products
    uint32_t inclusive;   // output includes element i itself
    uint32_t has_carry;   // 0: top level, bcarry is not read
    uint32_t fused_nb;    // phase 3, <= SCAN_T blocks: every workgroup scans the bpart[0..fused_nb) block totals itself instead of reading bcarry
    uint32_t* total_out;  // fused: where block 0 leaves the combination of everything
    Limbs9 zpow[20];      // MODE 1: z^(RUN * 2^j), j < 8, for the thread-level doubling (2^261 form); [8] = z, [9] = z^RUN;
                          // [12 + j] = z^(SCAN_BLOCK * 2^j), j < 8: the doubling over whole blocks of the fused phase 3
};

// LDS doubling scan over the SCAN_T thread partials of a block.  v = this thread's inclusive partial on entry; returns the
// exclusive partial (combination of all LATER threads for suffix scans / EARLIER threads for prefix scans).
template <int MODE, int ZB = 0> __device__ __forceinline__ FrM block_scan(FrM v, uint32_t* sh, bool towards_high, const ScanArgs& A, FrM* total)
{
    // position p runs in scan order: p = 0 is the first element combined
    const uint32_t t = threadIdx.x;
    const uint32_t p = towards_high ? t : (SCAN_T - 1 - t);
    // inclusive Hillis-Steele over positions: after step j, v_p = combine(v_{p - 2^j .. p})
    uint32_t* buf = sh;
#pragma unroll 1
    for (int j = 0, off = 1; off < SCAN_T; j++, off <<= 1) {
#pragma unroll
        for (int k = 0; k < NL; k++) buf[k * SCAN_T + p] = v.d[k];
        __syncthreads();
        if (p >= (uint32_t)off) {
            FrM o;
#pragma unroll
            for (int k = 0; k < NL; k++) o.d[k] = buf[k * SCAN_T + p - off];
            if (MODE == 0) {
                v = mulv(v, o);
            } else {
                // suffix Horner: scan order runs from the highest index down; position p holds S over `off` earlier
                // positions: S_new = S_here + z^(RUN * off) * S_earlier ... earlier positions are HIGHER indices, so
                // S(i..) = S_here + z^(len_here) * S_later with len_here = RUN * off thread-runs combined so far
                v = tight2<Fr>(add(v, mul(o, cst(A.zpow[ZB + j]))));
            }
        }
        __syncthreads();
    }
    // v is now inclusive over positions 0..p; the exclusive value is position p-1's inclusive value
#pragma unroll
    for (int k = 0; k < NL; k++) buf[k * SCAN_T + p] = v.d[k];
    __syncthreads();
    FrM ex;
    if (p == 0) {
        if (MODE == 0) ex = mul(fe_from<Fr>(Fr::ONE_M256), fe_from<Fr>(Fr::ONE)); // memory-form one
        else ex = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    } else {
#pragma unroll
        for (int k = 0; k < NL; k++) ex.d[k] = buf[k * SCAN_T + p - 1];
    }
    FrM tot;
#pragma unroll
    for (int k = 0; k < NL; k++) tot.d[k] = buf[k * SCAN_T + SCAN_T - 1];
    *total = tot;
    __syncthreads();
    return ex;
}

// identity-padded load: MODE 0 pads with one, MODE 1 with zero
template <int MODE> __device__ __forceinline__ FrV ld_or_id(const uint32_t* p, uint32_t i, uint32_t n)
{
    if (i < n) return ldv(p, i);
    FrV r;
    if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < NL; k++) r.d[k] = Fr::ONE_M256[k];
    } else {
#pragma unroll
        for (int k = 0; k < NL; k++) r.d[k] = 0;
    }
    return r;
}

// phase 1: per-thread run totals, block scan of them -> tpart (exclusive within the block), bpart (block total)
// (two independent scans per launch: blockIdx.y selects the argument block; the second may be empty, n = 0)
template <int MODE> __global__ void __launch_bounds__(SCAN_T) k_scan_phase1(ScanArgs A0, ScanArgs A1)
{
    __shared__ uint32_t sh[NL * SCAN_T];
    const ScanArgs& A = blockIdx.y ? A1 : A0;
    if (blockIdx.x * SCAN_BLOCK >= A.n) return; // whole workgroup: the two scans may differ in length
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint32_t base = b * SCAN_BLOCK + t * RUN;
    const bool suffix = (MODE == 1) || A.reverse;
    FrM run;
    if (MODE == 0) {
        FrV x0 = ld_or_id<0>(A.in, base, A.n);
        run = mul(x0, fe_from<Fr>(Fr::ONE)); // memory form kept
#pragma unroll
        for (int k = 1; k < RUN; k++) run = mulv(run, ld_or_id<0>(A.in, base + k, A.n));
    } else {
        // E = sum_k x_{base+k} z^k, Horner from the top
        const FrC z = cst(A.zpow[8]);
        FrH h = ld_or_id<1>(A.in, base + RUN - 1, A.n);
#pragma unroll
        for (int k = RUN - 2; k >= 0; k--) h = add(mul(h, z), ld_or_id<1>(A.in, base + k, A.n));
        run = tight2<Fr>(h);
    }
    FrM total;
    FrM ex = block_scan<MODE>(run, sh, !suffix, A, &total);
    const uint32_t tid = b * SCAN_T + t;
    if ((size_t)tid * RUN < A.n) stv(A.tpart, tid, ex);
    if (t == 0) stv(A.bpart, b, total);
}

// phase 3: out_i from the block carry, the thread partial and the in-run elements
template <int MODE> __global__ void __launch_bounds__(SCAN_T) k_scan_phase3(ScanArgs A0, ScanArgs A1)
{
    __shared__ uint32_t sh[NL * SCAN_T];
    const ScanArgs& A = blockIdx.y ? A1 : A0;
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    if (b * SCAN_BLOCK >= A.n) return; // whole workgroup: the two scans may differ in length
    const bool suffix = (MODE == 1) || A.reverse;
    // <= SCAN_T blocks: the exclusive scan of the block totals (phase 1 + phase 3 of the nested level: two launches of one workgroup
    // each, ~40 us of a 2^16-element scan's ~95) is done here by every workgroup for itself, lane t standing for block t
    FrM carry = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    if (A.fused_nb) {
        FrM v;
        if (MODE == 0) {
            v = mul(ld_or_id<0>(A.bpart, t, A.fused_nb), fe_from<Fr>(Fr::ONE));
        } else {
            FrH h = ld_or_id<1>(A.bpart, t, A.fused_nb);
            v = tight2<Fr>(h);
        }
        FrM total;
        const FrM ex = block_scan<MODE, 12>(v, sh, !suffix, A, &total);
        if (t == b) {
#pragma unroll
            for (int k = 0; k < NL; k++) sh[k] = ex.d[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NL; k++) carry.d[k] = sh[k];
        if (b == 0 && t == 0 && A.total_out) stv(A.total_out, 0, total);
    }
    const uint32_t base = b * SCAN_BLOCK + t * RUN;
    if (base >= A.n || !A.out) return; // (a scan wanted for its total only has no output vector)
    const uint32_t tid = b * SCAN_T + t;
    if (MODE == 0) {
        FrM acc = mul(ldv(A.tpart, tid), fe_from<Fr>(Fr::ONE)); // everything before (after) this run
        if (A.fused_nb) acc = mulv(carry, ldv(A.tpart, tid));
        else if (A.has_carry) acc = mulv(ldv(A.bcarry, b), ldv(A.tpart, tid));
        if (!suffix) {
            for (uint32_t k = 0; k < RUN && base + k < A.n; k++) {
                const FrV x = ldv(A.in, base + k);
                if (A.inclusive) {
                    acc = mulv(acc, x);
                    stv(A.out, base + k, acc);
                } else {
                    stv(A.out, base + k, acc);
                    acc = mulv(acc, x);
                }
            }
        } else {
            for (int k = RUN - 1; k >= 0; k--) {
                if (base + k >= A.n) continue;
                const FrV x = ldv(A.in, base + k);
                if (A.inclusive) {
                    acc = mulv(acc, x);
                    stv(A.out, base + k, acc);
                } else {
                    stv(A.out, base + k, acc);
                    acc = mulv(acc, x);
                }
            }
        }
    } else {
        // carry into this run: S at index base + RUN = tpart + z^(RUN * (threads after t in the block)) * bcarry
        const uint32_t after = SCAN_T - 1 - t;
        FrM zp = mul(fe_from<Fr>(Fr::ONE), fe_from<Fr>(Fr::ONE)); // one in the 2^261 form
        for (int j = 0; j < 8; j++)
            if ((after >> j) & 1) zp = mul(zp, cst(A.zpow[j]));
        FrH acc = ldv(A.tpart, tid);
        if (A.fused_nb) acc = add(mul(carry, zp), ldv(A.tpart, tid));
        else if (A.has_carry) acc = add(mul(ldv(A.bcarry, b), zp), ldv(A.tpart, tid));
        const FrC z = cst(A.zpow[8]);
        for (int k = RUN - 1; k >= 0; k--) {
            if (base + k >= A.n) continue; // padded zeros do not change acc
            const FrV x = ldv(A.in, base + k);
            if (A.inclusive) {
                acc = add(mul(acc, z), x);
                stv(A.out, base + k, acc);
            } else {
                stv(A.out, base + k, acc);
                acc = add(mul(acc, z), x);
            }
        }
    }
}

// ---- evaluate: sum_i c_i z^i ------------------------------------------------------------------------------------------
// thread t owns indices t, t + T, t + 2T, ...: Horner in z^T from the top, then * z^t; block tree in LDS; one partial / block
__global__ void __launch_bounds__(PT) k_eval_partial(const uint32_t* __restrict__ c, uint32_t n, PowTab T, Limbs9 zT_m261, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t sh[NL * PT];
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    FrM acc = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    if (t < n) {
        const uint32_t cnt = (n - t + nt - 1) / nt;
        const FrC zT = cst(zT_m261);
        FrH h = ldv(c, t + (size_t)(cnt - 1) * nt);
        for (uint32_t k = cnt - 1; k-- > 0;) h = add(mul(h, zT), ldv(c, t + (size_t)k * nt));
        Limbs9 one261;
#pragma unroll
        for (int k = 0; k < NL; k++) one261.d[k] = Fr::ONE[k];
        acc = mul(h, pow_tab(T, t, one261)); // * z^t
    }
    // block sum (lazy adds: 8 levels of doubling the bound stay far below the limits after a tighten per level)
#pragma unroll
    for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = acc.d[k];
    __syncthreads();
    for (uint32_t half = PT / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) {
            FrM a, b;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                a.d[k] = sh[k * PT + threadIdx.x];
                b.d[k] = sh[k * PT + threadIdx.x + half];
            }
            FrM s = tight2<Fr>(add(a, b));
#pragma unroll
            for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = s.d[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FrM s;
#pragma unroll
        for (int k = 0; k < NL; k++) s.d[k] = sh[k * PT];
        stv(partial, blockIdx.x, s);
    }
}
// out[0] = sum of `count` partials (count <= a few hundred): one workgroup
__global__ void __launch_bounds__(PT) k_sum_small(const uint32_t* __restrict__ partial, uint32_t count, uint32_t* __restrict__ out)
{
    __shared__ uint32_t sh[NL * PT];
    FrM acc = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    for (uint32_t i = threadIdx.x; i < count; i += PT) acc = tight2<Fr>(add(acc, ldv(partial, i)));
#pragma unroll
    for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = acc.d[k];
    __syncthreads();
    for (uint32_t half = PT / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) {
            FrM a, b;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                a.d[k] = sh[k * PT + threadIdx.x];
                b.d[k] = sh[k * PT + threadIdx.x + half];
            }
            FrM s = tight2<Fr>(add(a, b));
#pragma unroll
            for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = s.d[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FrM s;
#pragma unroll
        for (int k = 0; k < NL; k++) s.d[k] = sh[k * PT];
        stv(out, 0, s);
    }
}

// several evaluations in one pair of launches (blockIdx.y = job): the prover's seven openings at z / z w (prover.cpp:478-512)
__global__ void __launch_bounds__(PT) k_eval_partial_batch(EvalBatchArgs A)
{
    __shared__ uint32_t sh[NL * PT];
    const uint32_t job = blockIdx.y;
    const uint32_t n = A.n[job], nblocks = A.blocks[job];
    if (blockIdx.x >= nblocks) return;
    const uint32_t* __restrict__ c = A.c[job];
    const PowTab& T = A.T[A.zsel[job]];
    const uint32_t nt = nblocks * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    FrM acc = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    if (t < n) {
        const uint32_t cnt = (n - t + nt - 1) / nt;
        const FrC zT = cst(A.zT[job]);
        FrH h = ldv(c, t + (size_t)(cnt - 1) * nt);
        for (uint32_t k = cnt - 1; k-- > 0;) h = add(mul(h, zT), ldv(c, t + (size_t)k * nt));
        Limbs9 one261;
#pragma unroll
        for (int k = 0; k < NL; k++) one261.d[k] = Fr::ONE[k];
        acc = mul(h, pow_tab(T, t, one261)); // * z^t
    }
#pragma unroll
    for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = acc.d[k];
    __syncthreads();
    for (uint32_t half = PT / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) {
            FrM a, b;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                a.d[k] = sh[k * PT + threadIdx.x];
                b.d[k] = sh[k * PT + threadIdx.x + half];
            }
            FrM sm = tight2<Fr>(add(a, b));
#pragma unroll
            for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = sm.d[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FrM sm;
#pragma unroll
        for (int k = 0; k < NL; k++) sm.d[k] = sh[k * PT];
        stv(A.partial, (size_t)job * 256 + blockIdx.x, sm);
    }
}
__global__ void __launch_bounds__(PT) k_sum_small_batch(EvalBatchArgs A)
{
    __shared__ uint32_t sh[NL * PT];
    const uint32_t job = blockIdx.x, count = A.blocks[job];
    const uint32_t* partial = A.partial + (size_t)job * 256 * 8;
    FrM acc = mul(fe_zero<Fr>(), fe_from<Fr>(Fr::ONE));
    for (uint32_t i = threadIdx.x; i < count; i += PT) acc = tight2<Fr>(add(acc, ldv(partial, i)));
#pragma unroll
    for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = acc.d[k];
    __syncthreads();
    for (uint32_t half = PT / 2; half >= 1; half >>= 1) {
        if (threadIdx.x < half) {
            FrM a, b;
#pragma unroll
            for (int k = 0; k < NL; k++) {
                a.d[k] = sh[k * PT + threadIdx.x];
                b.d[k] = sh[k * PT + threadIdx.x + half];
            }
            FrM sm = tight2<Fr>(add(a, b));
#pragma unroll
            for (int k = 0; k < NL; k++) sh[k * PT + threadIdx.x] = sm.d[k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        FrM sm;
#pragma unroll
        for (int k = 0; k < NL; k++) sm.d[k] = sh[k * PT];
        stv(A.result[job], 0, sm);
    }
}

// ---- prover round kernels -------------------------------------------------------------------------------------------
// permutation.hpp:15-87: sigma[i] = k_type * w^(mapping & mask), gathered from the table of subgroup elements
__global__ void __launch_bounds__(PT) k_sigma_from_mapping(uint32_t* __restrict__ out, const uint32_t* __restrict__ mapping,
                                                         const uint32_t* __restrict__ roots, uint32_t n, Limbs9 k1_m261, Limbs9 k2_m261)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = mapping[i];
    const uint32_t idx = (m & ((1u << 29) - 1u)) & (n - 1);
    const FrV w = ldv(roots, idx);
    switch ((m >> 30) & 3u) {
    case 2: stv(out, i, mul(w, cst(k2_m261))); break;
    case 1: stv(out, i, mul(w, cst(k1_m261))); break;
    default: stv(out, i, w); break;
    }
}

// prover.cpp:148-187: numerator / denominator factors of the grand product, three wires multiplied together
//   num_i = (w_l + beta w^i + gamma)(w_r + beta k1 w^i + gamma)(w_o + beta k2 w^i + gamma)
//   den_i = (w_l + beta sigma_1 + gamma)(w_r + beta sigma_2 + gamma)(w_o + beta sigma_3 + gamma)
__global__ void __launch_bounds__(PT) k_z_terms(ZTermsArgs A)
{
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n) return;
    Limbs9 one261;
#pragma unroll
    for (int k = 0; k < NL; k++) one261.d[k] = Fr::ONE[k];
    FrM x = pow_tab(A.root, t, one261); // w^t, 2^261 form
    const FrC step = cst(A.step_m261), beta = cst(A.beta_m256), bk1 = cst(A.beta_k1_m256), bk2 = cst(A.beta_k2_m256), gamma = cst(A.gamma_m256),
              beta261 = cst(A.beta_m261);
    for (uint32_t i = t; i < A.n; i += nt) {
        const FrV wl = ldv(A.w_l, i), wr = ldv(A.w_r, i), wo = ldv(A.w_o, i);
        auto a0 = add(add(mul(x, beta), gamma), wl);
        auto a1 = add(add(mul(x, bk1), gamma), wr);
        auto a2 = add(add(mul(x, bk2), gamma), wo);
        stv(A.num, i, mul(mul(mul(a0, a1), a2), fix3()));
        auto b0 = add(add(mul(ldv(A.s1, i), beta261), gamma), wl);
        auto b1 = add(add(mul(ldv(A.s2, i), beta261), gamma), wr);
        auto b2 = add(add(mul(ldv(A.s3, i), beta261), gamma), wo);
        stv(A.den, i, mul(mul(mul(b0, b1), b2), fix3()));
        x = mul(x, step);
    }
}

// prover.cpp:253-269: dst (4n) = beta sigma(X) + w(X) + gamma in coefficient form, zero-padded
__global__ void __launch_bounds__(PT) k_sigma_prepare(uint32_t* __restrict__ dst, const uint32_t* __restrict__ sigma, const uint32_t* __restrict__ w,
                                                    uint32_t n, uint32_t n_dst, Limbs9 gamma_m256)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_dst) return;
    if (i >= n) {
        uint4* q = reinterpret_cast<uint4*>(dst + (size_t)i * 8);
        q[0] = make_uint4(0, 0, 0, 0);
        q[1] = make_uint4(0, 0, 0, 0);
        return;
    }
    auto v = add(ldv(sigma, i), ldv(w, i));
    if (i == 0) stv(dst, i, add(v, cst(gamma_m256)));
    else stv(dst, i, v);
}

// prover.cpp:294-299 and :310-341 fused: the degree-3n part of the quotient numerator on the 4n coset
//   q[i] = (w_l + beta x + gamma)(w_r + beta k1 x + gamma)(w_o + beta k2 x + gamma) zf[i]  -  s1 s2 s3 zf[i + 4],   x = g w_4n^i
// (zf = alpha * Z on the coset, index i + 4 wraps: prover.cpp:286-289 appends the first four values instead)
__global__ void __launch_bounds__(PT) k_quotient_large(QuotLargeArgs A)
{
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= A.n4) return;
    FrM x = pow_tab(A.root, t, A.g_m261);
    const FrC step = cst(A.step_m261), beta = cst(A.beta_m256), bk1 = cst(A.beta_k1_m256), bk2 = cst(A.beta_k2_m256), gamma = cst(A.gamma_m256);
    for (uint32_t i = t; i < A.n4; i += nt) {
        auto t0 = add(add(mul(x, beta), gamma), ldv(A.wl_f, i));
        auto t1 = add(add(mul(x, bk1), gamma), ldv(A.wr_f, i));
        auto t2 = add(add(mul(x, bk2), gamma), ldv(A.wo_f, i));
        FrM id = mul(mul(mul(mul(t0, t1), t2), ldv(A.z_f, i)), fix4());
        FrM pm = mul(mul(mul(mul(ldv(A.s1_f, i), ldv(A.s2_f, i)), ldv(A.s3_f, i)), ldv(A.z_f, (i + 4) & (A.n4 - 1))), fix4());
        stv(A.q, i, sub(id, pm));
        x = mul(x, step);
    }
}

// prover.cpp:360-402 and arithmetic_widget.cpp:86-101 fused: the degree-2n part on the 2n coset
//   q[i] = (zf[2i+4] - alpha) alpha l1[i+4] + (zf[2i] - alpha) alpha^2 l1[i]
//        + abase (qm wl wr + ql wl + qr wr + qo wo + qc),  wires at index 2i of their 4n transforms
__global__ void __launch_bounds__(PT) k_quotient_mid(QuotMidArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n2) return;
    const uint32_t m4 = 2 * A.n2 - 1, m2 = A.n2 - 1;
    const FrC alpha = cst(A.alpha_m256);
    auto t6 = mul(mul(sub(ldv(A.z_f, (2 * i + 4) & m4), alpha), ldv(A.l1, (i + 4) & m2)), cst(A.alpha_fix_m261));   // * alpha * 2^5
    auto t4 = mul(mul(sub(ldv(A.z_f, 2 * i), alpha), ldv(A.l1, i)), cst(A.alpha2_fix_m261));                          // * alpha^2 * 2^5
    const FrV wl = ldv(A.wl_f, 2 * i), wr = ldv(A.wr_f, 2 * i), wo = ldv(A.wo_f, 2 * i);
    // selector transforms are stored unscaled; abase carries the fix factors: three-operand term needs 2^10, two-operand 2^5
    auto g3 = mul(mul(mul(ldv(A.qm_f, i), wl), wr), cst(A.abase_fix3_m261));
    auto gl = mul(ldv(A.ql_f, i), wl);
    auto gr = mul(ldv(A.qr_f, i), wr);
    auto go = mul(ldv(A.qo_f, i), wo);
    auto g2 = mul(add(add(gl, gr), go), cst(A.abase_fix2_m261));
    auto gc = mul(ldv(A.qc_f, i), cst(A.abase_m261));
    stv(A.q, i, add(add(add(t6, t4), add(g3, g2)), gc));
}

// mimc_widget.cpp:58-90 on the 4n coset: with T0 = w_o + w_l + q_coef,
//   q[i] += abase q_sel [ (T0^3 - w_r) + alpha (w_r^2 T0 - w_o[i + 4]) ]        (w_o[i + 4] = the next gate's output wire)
__global__ void __launch_bounds__(PT) k_quotient_mimc(QuotMimcArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n4) return;
    const FrV wl = ldv(A.wl_f, i), wr = ldv(A.wr_f, i), wo = ldv(A.wo_f, i), won = ldv(A.wo_f, (i + 4) & (A.n4 - 1));
    const auto t0 = weak(add(add(wo, wl), ldv(A.qcoef_f, i)));
    const auto t0sq = mulv(t0, t0);
    const auto t1 = weak(sub(mulv(t0sq, t0), wr));
    const auto t2 = mul(weak(sub(mulv(mulv(wr, wr), t0), won)), cst(A.alpha_m261));
    const auto sum = weak(add(t1, t2));
    stv(A.q, i, add(mul(mul(sum, ldv(A.qsel_f, i)), cst(A.abase_fix_m261)), ldv(A.q, i)));
}

// sequential_widget.cpp:47-62: q[i] += c q_o_next[i] w_o[2i + 4] on the 2n coset (index 2i + 4 of the 4n evaluations = the next gate's row)
__global__ void __launch_bounds__(PT) k_quotient_seq(QuotSeqArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n2) return;
    const uint32_t m4 = 2 * A.n2 - 1;
    auto t = mul(mul(ldv(A.qon_f, i), ldv(A.wo_f, (2 * i + 4) & m4)), cst(A.c_fix_m261));
    stv(A.q, i, add(t, ldv(A.q, i)));
}

// bool_widget.cpp:62-100: q[i] += c_l q_bl (w_l^2 - w_l) + c_r q_br (w_r^2 - w_r) + c_o q_bo (w_o^2 - w_o), wires at index 2i
__global__ void __launch_bounds__(PT) k_quotient_bool(QuotBoolArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n2) return;
    const FrV wl = ldv(A.wl_f, 2 * i), wr = ldv(A.wr_f, 2 * i), wo = ldv(A.wo_f, 2 * i);
    auto tl = mul(mul(weak(sub(mulv(wl, wl), wl)), ldv(A.qbl_f, i)), cst(A.cl_fix_m261));
    auto tr = mul(mul(weak(sub(mulv(wr, wr), wr)), ldv(A.qbr_f, i)), cst(A.cr_fix_m261));
    auto to = mul(mul(weak(sub(mulv(wo, wo), wo)), ldv(A.qbo_f, i)), cst(A.co_fix_m261));
    stv(A.q, i, add(add(add(tl, tr), to), ldv(A.q, i)));
}

// polynomial_arithmetic.cpp:478-560: c[i] *= (x_i - w_n^-1) / ((x_i)^n - 1),  x_i = g w_N^i;  (x_i)^n - 1 takes k = N/n values
__global__ void __launch_bounds__(PT) k_divide_vanishing(uint32_t* __restrict__ c, uint32_t N, uint32_t k, PowTab root, Limbs9 g_m261, Limbs9 step_m261,
                                                       Limbs9 wninv_m261, Limbs9 inv0, Limbs9 inv1, Limbs9 inv2, Limbs9 inv3)
{
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    FrM x = pow_tab(root, t, g_m261);
    const FrC step = cst(step_m261), wninv = cst(wninv_m261);
    const uint32_t sel = t & (k - 1); // nt is a multiple of 4, so i mod k is fixed per thread
    const FrC iv = cst(sel == 0 ? inv0 : sel == 1 ? inv1 : sel == 2 ? inv2 : inv3);
    for (uint32_t i = t; i < N; i += nt) {
        auto num = weak(sub(x, wninv));                 // 2^261 form
        stv(c, i, mul(mul(ldv(c, i), num), iv));
        x = mul(x, step);
    }
}

// polynomial_arithmetic.cpp:381-476, first half: d[i] = g w_N^i - 1 (to be inverted)
__global__ void __launch_bounds__(PT) k_l1_denominators(uint32_t* __restrict__ d, uint32_t N, PowTab root, Limbs9 g_m256, Limbs9 step_m261)
{
    const uint32_t nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    FrM x = pow_tab(root, t, g_m256); // memory form
    const FrC step = cst(step_m261), one = fe_from<Fr>(Fr::ONE_M256);
    for (uint32_t i = t; i < N; i += nt) {
        stv(d, i, sub(x, one));
        x = mul(x, step);
    }
}
// second half: l1[i] = inv[i] * ((g^n w_k^(i mod k)) - 1) / n
__global__ void __launch_bounds__(PT) k_l1_scale(uint32_t* __restrict__ l1, uint32_t N, uint32_t k, Limbs9 s0, Limbs9 s1, Limbs9 s2, Limbs9 s3)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t sel = i & (k - 1);
    FrC s;
#pragma unroll
    for (int j = 0; j < NL; j++) s.d[j] = sel == 0 ? s0.d[j] : sel == 1 ? s1.d[j] : sel == 2 ? s2.d[j] : s3.d[j];
    stv(l1, i, mul(ldv(l1, i), s));
}

// prover.cpp:520-528 + arithmetic_widget.cpp:106-126: r[i] = sum_j c_j p_j[i] over the seven coefficient-form polynomials
__global__ void __launch_bounds__(PT) k_lincomb(LinCombArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    FrM acc = mul(ldv(A.p[0], i), cst(A.c[0]));
    for (int j = 1; j < A.count; j++) acc = tight2<Fr>(add(acc, mul(ldv(A.p[j], i), cst(A.c[j]))));
    if (A.out_add) acc = tight2<Fr>(add(acc, ldv(A.out_add, i)));
    stv(A.out, i, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static inline uint32_t pw_blocks(size_t n) { return (uint32_t)((n + PT - 1) / PT); }
// grid for the strided kernels: a multiple of 4 threads in total, at most 1024 workgroups
static inline uint32_t strided_blocks(size_t n)
{
    size_t b = (n + 4 * PT - 1) / (4 * PT);
    return (uint32_t)std::min<size_t>(std::max<size_t>(b, 1), 1024);
}

PowTab make_powtab(const host::Fr& base)
{
    PowTab T;
    host::Fr b = base;
    for (int j = 0; j < 24; j++) {
        T.p[j] = host::limbs_m261(b);
        b = host::fr_sqr(b);
    }
    return T;
}

int Scratch::ensure(size_t bytes)
{
    if (bytes <= cap) return BBGPU_OK;
    if (base) (void)dev_free(base);
    base = nullptr;
    cap = 0;
    HIPCHK(dev_malloc((void**)&base, bytes));
    cap = bytes;
    return BBGPU_OK;
}
void Scratch::release()
{
    if (base) (void)dev_free(base);
    if (h_pinned) (void)hipHostFree(h_pinned);
    base = nullptr;
    h_pinned = nullptr;
    cap = 0;
}
static int ensure_pinned(Scratch& S)
{
    if (!S.h_pinned) HIPCHK(hipHostMalloc((void**)&S.h_pinned, 4096));
    return BBGPU_OK;
}

int powers(uint64_t* d_out, size_t n, const host::Fr& base, const host::Fr& start, hipStream_t st)
{
    const uint32_t blocks = strided_blocks(n);
    k_powers<<<blocks, PT, 0, st>>>((uint32_t*)d_out, (uint32_t)n, make_powtab(base), host::limbs_m256(start),
                                   host::limbs_m261(host::fr_pow(base, (uint64_t)blocks * PT)));
    HIPCHK(launch_check());
    return BBGPU_OK;
}
int copy_pad(uint64_t* d_dst, const uint64_t* d_src, size_t n_src, size_t n_dst, hipStream_t st)
{
    k_copy_pad<<<pw_blocks(n_dst), PT, 0, st>>>((uint32_t*)d_dst, (const uint32_t*)d_src, (uint32_t)n_src, (uint32_t)n_dst);
    HIPCHK(launch_check());
    return BBGPU_OK;
}
int add_inplace(uint64_t* d_a, const uint64_t* d_b, size_t n, hipStream_t st)
{
    k_add_inplace<<<pw_blocks(n), PT, 0, st>>>((uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t)n);
    HIPCHK(launch_check());
    return BBGPU_OK;
}
int mul_pointwise(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, hipStream_t st)
{
    k_mul<<<pw_blocks(n), PT, 0, st>>>((uint32_t*)d_out, (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t)n);
    HIPCHK(launch_check());
    return BBGPU_OK;
}
int mul2c(uint64_t* d_out, const uint64_t* d_a, const uint64_t* d_b, size_t n, const host::Fr& c, hipStream_t st)
{
    const host::Fr cf = host::fr_mul(c, host::fr_from_u64(32)); // c * 2^5: the FIX2 factor folded in
    k_mul2c<<<pw_blocks(n), PT, 0, st>>>((uint32_t*)d_out, (const uint32_t*)d_a, (const uint32_t*)d_b, (uint32_t)n, host::limbs_m261(cf));
    HIPCHK(launch_check());
    return BBGPU_OK;
}

// scratch layout of a scan over n elements: tpart, bpart, bcarry, plus the nested single-block scan's own tpart/bpart -- and, above 2^22 elements,
// where the scan over the block totals is a two-level scan of its own, that scan's scratch behind it
static size_t scan_scratch_own(size_t n)
{
    const size_t nt = (n + RUN - 1) / RUN, nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return (((nt + 2 * nb + (nb + RUN - 1) / RUN + 8) * 32 + 1024) + 255) & ~(size_t)255;
}
size_t scan_scratch_bytes(size_t n)
{
    const size_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    return scan_scratch_own(n) + (nb > (size_t)SCAN_BLOCK ? scan_scratch_bytes(nb) : 0);
}

// MODE 0 product scan (z ignored) / MODE 1 Horner suffix sums with multiplier z.  Three phases: runs of RUN per thread and a scan of the thread partials
// in LDS (phase 1), a scan over the block totals, the carry pass (phase 3).  The scan over the block totals is one workgroup up to 2^22 elements
// and the same three phases again above (round 4: n up to 2^28, the transforms' limit; the prover's L_1 scan is 2n).
// Up to two independent scans of the same mode share the launches (ScanJob[2]; the second may have n = 0).
static void scan_fill(int mode, const ScanJob& J, uint8_t* base, ScanArgs& A, ScanArgs& B)
{
    const size_t n = J.n;
    const size_t nt = (n + RUN - 1) / RUN, nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK, nt2 = (nb + RUN - 1) / RUN;
    uint32_t* tpart = (uint32_t*)base;
    uint32_t* bpart = tpart + nt * 8;
    uint32_t* bcarry = bpart + nb * 8;
    uint32_t* tpart2 = bcarry + nb * 8;
    uint32_t* bpart2 = tpart2 + nt2 * 8; // one element: the grand total
    A = ScanArgs{};
    A.in = (const uint32_t*)J.in;
    A.out = (uint32_t*)J.out;
    A.tpart = tpart;
    A.bpart = bpart;
    A.bcarry = bcarry;
    A.n = (uint32_t)n;
    A.reverse = J.reverse ? 1u : 0u;
    A.inclusive = J.inclusive ? 1u : 0u;
    A.has_carry = 1;
    B = A; // the nested scan over the block totals: exclusive, same direction
    B.in = bpart;
    B.out = bcarry;
    B.tpart = tpart2;
    B.bpart = bpart2;
    B.bcarry = nullptr;
    B.has_carry = 0;
    B.n = (uint32_t)nb;
    B.inclusive = 0;
    if (mode == 1) {
        host::Fr zr = host::fr_pow(J.z, RUN);
        host::Fr p = zr;
        for (int j = 0; j < 8; j++) { A.zpow[j] = host::limbs_m261(p); p = host::fr_sqr(p); }
        A.zpow[8] = host::limbs_m261(J.z);
        A.zpow[9] = host::limbs_m261(zr);
        // nested level: its "elements" are whole blocks, so its z is z^SCAN_BLOCK
        host::Fr zb = host::fr_pow(J.z, SCAN_BLOCK);
        p = zb;
        for (int j = 0; j < 8; j++) { A.zpow[12 + j] = host::limbs_m261(p); p = host::fr_sqr(p); }
        host::Fr zbr = host::fr_pow(zb, RUN);
        p = zbr;
        for (int j = 0; j < 8; j++) { B.zpow[j] = host::limbs_m261(p); p = host::fr_sqr(p); }
        B.zpow[8] = host::limbs_m261(zb);
        B.zpow[9] = host::limbs_m261(zbr);
    }
}
static int scan_pair_at(int mode, const ScanJob* jobs, int count, uint8_t* base, hipStream_t st);
int scan_pair(int mode, const ScanJob* jobs, int count, Scratch& S, hipStream_t st)
{
    if (count < 1 || count > 2) return BBGPU_ERR_ARG;
    size_t total = 0;
    for (int j = 0; j < count; j++) {
        if (jobs[j].n > ((size_t)1 << 28)) {
            set_error("scan of %zu elements: at most 2^28", jobs[j].n);
            return BBGPU_ERR_SIZE;
        }
        total += scan_scratch_bytes(jobs[j].n);
    }
    int rc = S.ensure(total + 64); // the whole nest at once: the workspace must not move under the launches of an outer level
    if (rc) return rc;
    return scan_pair_at(mode, jobs, count, S.base, st);
}
static int scan_pair_at(int mode, const ScanJob* jobs, int count, uint8_t* base, hipStream_t st)
{
    size_t off[3] = { 0, 0, 0 }, nbmax = 0;
    for (int j = 0; j < count; j++) {
        off[j + 1] = off[j] + scan_scratch_bytes(jobs[j].n);
        nbmax = std::max(nbmax, (jobs[j].n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    }
    if (nbmax == 0) return BBGPU_OK;
    struct { uint8_t* base; } S{ base };
    ScanArgs A[2], B[2];
    A[1] = ScanArgs{};
    B[1] = ScanArgs{};
    bool any_out = false;
    for (int j = 0; j < count; j++) {
        scan_fill(mode, jobs[j], S.base + off[j], A[j], B[j]);
        any_out = any_out || jobs[j].out;
    }
    ScanArgs A3[2] = { A[0], A[1] };
    for (int j = 0; j < count; j++)
        if (!jobs[j].out) A3[j].n = 0;
    const dim3 g1((uint32_t)nbmax, count), gb(1, count);
    static const bool fuse_env = [] { const char* e = getenv("BBGPU_SCAN_FUSED"); return !e || atoi(e) != 0; }(); // tuning knob
    const bool fused = fuse_env && nbmax <= (size_t)SCAN_T;
    if (fused) {
        for (int j = 0; j < count; j++) {
            A3[j].fused_nb = (uint32_t)((jobs[j].n + SCAN_BLOCK - 1) / SCAN_BLOCK);
            A3[j].total_out = B[j].bpart;
            A3[j].n = A[j].n; // also the scans without an output vector run phase 3 (for the total)
        }
        if (mode == 0) {
            k_scan_phase1<0><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
            k_scan_phase3<0><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
        } else {
            k_scan_phase1<1><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
            k_scan_phase3<1><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
        }
    } else
    if (nbmax > (size_t)SCAN_BLOCK) {
        // more block totals than one workgroup scans: the scan over them (exclusive, same direction; Horner: multiplier z^SCAN_BLOCK) is a scan of this
        // kind itself, run on the scratch behind this level's; its grand total lands where the one-workgroup form leaves it (B.bpart)
        if (mode == 0) k_scan_phase1<0><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
        else k_scan_phase1<1><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
        for (int j = 0; j < count; j++) {
            const size_t nb = (jobs[j].n + SCAN_BLOCK - 1) / SCAN_BLOCK;
            if (nb == 0) continue;
            if (nb <= (size_t)SCAN_BLOCK) { // the shorter of two jobs may still fit a workgroup: the one-block form, alone in its launch
                ScanArgs none{};
                if (mode == 0) {
                    k_scan_phase1<0><<<dim3(1, 1), SCAN_T, 0, st>>>(B[j], none);
                    k_scan_phase3<0><<<dim3(1, 1), SCAN_T, 0, st>>>(B[j], none);
                } else {
                    k_scan_phase1<1><<<dim3(1, 1), SCAN_T, 0, st>>>(B[j], none);
                    k_scan_phase3<1><<<dim3(1, 1), SCAN_T, 0, st>>>(B[j], none);
                }
                continue;
            }
            ScanJob I{};
            I.in = (const uint64_t*)A[j].bpart;
            I.out = (uint64_t*)A[j].bcarry;
            I.n = nb;
            I.reverse = jobs[j].reverse;
            I.inclusive = false;
            I.d_total = (uint64_t*)B[j].bpart;
            if (mode == 1) I.z = host::fr_pow(jobs[j].z, SCAN_BLOCK);
            if (int rc = scan_pair_at(mode, &I, 1, S.base + off[j] + scan_scratch_own(jobs[j].n), st)) return rc;
        }
        if (any_out) {
            if (mode == 0) k_scan_phase3<0><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
            else k_scan_phase3<1><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
        }
    } else
    if (mode == 0) {
        k_scan_phase1<0><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
        k_scan_phase1<0><<<gb, SCAN_T, 0, st>>>(B[0], B[1]);
        k_scan_phase3<0><<<gb, SCAN_T, 0, st>>>(B[0], B[1]);
        if (any_out) k_scan_phase3<0><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
    } else {
        k_scan_phase1<1><<<g1, SCAN_T, 0, st>>>(A[0], A[1]);
        k_scan_phase1<1><<<gb, SCAN_T, 0, st>>>(B[0], B[1]);
        k_scan_phase3<1><<<gb, SCAN_T, 0, st>>>(B[0], B[1]);
        if (any_out) k_scan_phase3<1><<<g1, SCAN_T, 0, st>>>(A3[0], A3[1]);
    }
    HIPCHK(launch_check());
    for (int j = 0; j < count; j++)
        if (jobs[j].d_total && jobs[j].n) HIPCHK(hipMemcpyAsync(jobs[j].d_total, B[j].bpart, 32, hipMemcpyDeviceToDevice, st));
    return BBGPU_OK;
}
static int scan_run(int mode, const uint64_t* d_in, uint64_t* d_out, size_t n, bool reverse, bool inclusive, const host::Fr* z, Scratch& S,
                    hipStream_t st, uint64_t* d_total /* optional: 32-byte device slot receiving the full combination */)
{
    if (n == 0) return BBGPU_OK;
    ScanJob J{};
    J.in = d_in; J.out = d_out; J.n = n; J.reverse = reverse; J.inclusive = inclusive; J.d_total = d_total;
    if (z) J.z = *z;
    return scan_pair(mode, &J, 1, S, st);
}

int product_scan(const uint64_t* d_in, uint64_t* d_out, size_t n, bool reverse, bool inclusive, Scratch& S, hipStream_t st, uint64_t* d_total)
{
    return scan_run(0, d_in, d_out, n, reverse, inclusive, nullptr, S, st, d_total);
}
int horner_suffix(const uint64_t* d_in, uint64_t* d_out, size_t n, const host::Fr& z, bool inclusive, Scratch& S, hipStream_t st, uint64_t* d_total)
{
    return scan_run(1, d_in, d_out, n, true, inclusive, &z, S, st, d_total);
}

// polynomial_arithmetic::evaluate (:337-373): result left in a 32-byte device slot (canonical); evaluate() also fetches it
// workgroups of an evaluation: every lane pays ~log2(lanes) multiplications for its z^t on top of one per coefficient, so a lane takes
// 8 coefficients (Horner in z^T) before the grid grows: at n = 2^16 one lane per coefficient cost 17 dependent multiplications each
// (the prover's seven openings: 73 us), 8 per lane cost 8 + 13 for eight (21 us)
static uint32_t eval_blocks(size_t n)
{
    return (uint32_t)std::min<size_t>(std::max<size_t>((n + (size_t)PT * 8 - 1) / ((size_t)PT * 8), 1), 256);
}
int evaluate_to_device(const uint64_t* d_coeffs, size_t n, const host::Fr& z, uint64_t* d_result, Scratch& S, hipStream_t st)
{
    if (n == 0) {
        HIPCHK(hipMemsetAsync(d_result, 0, 32, st));
        return BBGPU_OK;
    }
    const uint32_t blocks = eval_blocks(n);
    int rc = S.ensure((size_t)blocks * 32 + 64);
    if (rc) return rc;
    const uint32_t nt = blocks * PT;
    k_eval_partial<<<blocks, PT, 0, st>>>((const uint32_t*)d_coeffs, (uint32_t)n, make_powtab(z), host::limbs_m261(host::fr_pow(z, nt)), (uint32_t*)S.base);
    k_sum_small<<<1, PT, 0, st>>>((const uint32_t*)S.base, blocks, (uint32_t*)d_result);
    HIPCHK(launch_check());
    return BBGPU_OK;
}
int evaluate(const uint64_t* d_coeffs, size_t n, const host::Fr& z, host::Fr* out, Scratch& S, hipStream_t st)
{
    int rc = ensure_pinned(S);
    if (rc) return rc;
    rc = S.ensure(((n + PT - 1) / PT + 8) * 32 + 128);
    if (rc) return rc;
    // the partials live at the start of the scratch; the result slot sits past them
    const size_t slot_off = (std::min<size_t>((n + PT - 1) / PT, 256) + 1) * 32;
    uint64_t* d_res = (uint64_t*)(S.base + slot_off);
    rc = evaluate_to_device(d_coeffs, n, z, d_res, S, st);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(S.h_pinned, d_res, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(out->d, S.h_pinned, 32);
    return BBGPU_OK;
}

// up to 10 evaluations (each at z[0] or z[1]) in one pair of launches; results land in the jobs' 32-byte device slots
int evaluate_batch_to_device(const EvalJob* jobs, int count, const host::Fr z[2], Scratch& S, hipStream_t st)
{
    if (count < 1 || count > 10) return BBGPU_ERR_ARG;
    int rc = S.ensure((size_t)10 * 256 * 32 + 64);
    if (rc) return rc;
    EvalBatchArgs A{};
    A.T[0] = make_powtab(z[0]);
    A.T[1] = make_powtab(z[1]);
    A.partial = (uint32_t*)S.base;
    uint32_t maxb = 1;
    for (int j = 0; j < count; j++) {
        const size_t n = jobs[j].n;
        const uint32_t blocks = eval_blocks(n);
        A.c[j] = (const uint32_t*)jobs[j].coeffs;
        A.n[j] = (uint32_t)n;
        A.blocks[j] = blocks;
        A.zsel[j] = (uint8_t)(jobs[j].zsel ? 1 : 0);
        A.zT[j] = host::limbs_m261(host::fr_pow(z[A.zsel[j]], (uint64_t)blocks * PT));
        A.result[j] = (uint32_t*)jobs[j].d_result;
        maxb = std::max(maxb, blocks);
    }
    k_eval_partial_batch<<<dim3(maxb, count), PT, 0, st>>>(A);
    k_sum_small_batch<<<count, PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

// fr::batch_invert (field.hpp:503-522) on a resident vector: inv(a_i) = (prod_{j<i} a_j)(prod_{j>i} a_j) / prod_j a_j.
// Two product scans and ONE field inversion on the host (10 us; a Fermat chain on one GPU lane takes ~0.25 ms).
// d_tmp: n elements of workspace.  All a_i must be non-zero (the reference's loop has the same precondition).
int batch_invert(uint64_t* d_v, uint64_t* d_tmp, size_t n, Scratch& S, hipStream_t st)
{
    if (n == 0) return BBGPU_OK;
    int rc = ensure_pinned(S);
    if (rc) return rc;
    const size_t off_suffix = (scan_scratch_bytes(n) + 64 + 63) & ~(size_t)63;
    rc = S.ensure(off_suffix + n * 32 + 64);
    if (rc) return rc;
    uint64_t* d_suffix = (uint64_t*)(S.base + off_suffix);
    uint64_t* d_total = d_suffix + n * 4; // 32 bytes past the suffix array
    rc = product_scan(d_v, d_tmp, n, false, false, S, st, d_total); // exclusive prefix products -> d_tmp, total
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(S.h_pinned, d_total, 32, hipMemcpyDeviceToHost, st));
    rc = product_scan(d_v, d_suffix, n, true, false, S, st, nullptr); // exclusive suffix products
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    host::Fr total;
    memcpy(total.d, S.h_pinned, 32);
    rc = mul2c(d_v, d_tmp, d_suffix, n, host::fr_inv(total), st);
    return rc;
}

int sigma_from_mapping(uint64_t* d_out, const uint32_t* d_mapping, const uint64_t* d_roots, size_t n, hipStream_t st)
{
    k_sigma_from_mapping<<<pw_blocks(n), PT, 0, st>>>((uint32_t*)d_out, d_mapping, (const uint32_t*)d_roots, (uint32_t)n,
                                                     host::limbs_m261(host::fr_from_limbs(FrHostP::GEN5)), host::limbs_m261(host::fr_from_limbs(FrHostP::GEN7)));
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int z_terms(ZTermsArgs A, const host::Fr& root, const host::Fr& beta, const host::Fr& gamma, hipStream_t st)
{
    const uint32_t blocks = strided_blocks(A.n);
    A.root = make_powtab(root);
    A.step_m261 = host::limbs_m261(host::fr_pow(root, (uint64_t)blocks * PT));
    A.beta_m256 = host::limbs_m256(beta);
    A.beta_m261 = host::limbs_m261(beta);
    A.beta_k1_m256 = host::limbs_m256(host::fr_mul(beta, host::fr_from_limbs(FrHostP::GEN5)));
    A.beta_k2_m256 = host::limbs_m256(host::fr_mul(beta, host::fr_from_limbs(FrHostP::GEN7)));
    A.gamma_m256 = host::limbs_m256(gamma);
    k_z_terms<<<blocks, PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int sigma_prepare(uint64_t* d_dst, const uint64_t* d_sigma, const uint64_t* d_w, size_t n, size_t n_dst, const host::Fr& gamma, hipStream_t st)
{
    k_sigma_prepare<<<pw_blocks(n_dst), PT, 0, st>>>((uint32_t*)d_dst, (const uint32_t*)d_sigma, (const uint32_t*)d_w, (uint32_t)n, (uint32_t)n_dst,
                                                    host::limbs_m256(gamma));
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int quotient_large(QuotLargeArgs A, const host::Fr& root4n, const host::Fr& beta, const host::Fr& gamma, hipStream_t st)
{
    const uint32_t blocks = strided_blocks(A.n4);
    A.root = make_powtab(root4n);
    A.g_m261 = host::limbs_m261(host::fr_from_limbs(FrHostP::GEN5));
    A.step_m261 = host::limbs_m261(host::fr_pow(root4n, (uint64_t)blocks * PT));
    A.beta_m256 = host::limbs_m256(beta);
    A.beta_k1_m256 = host::limbs_m256(host::fr_mul(beta, host::fr_from_limbs(FrHostP::GEN5)));
    A.beta_k2_m256 = host::limbs_m256(host::fr_mul(beta, host::fr_from_limbs(FrHostP::GEN7)));
    A.gamma_m256 = host::limbs_m256(gamma);
    k_quotient_large<<<blocks, PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int quotient_mid(QuotMidArgs A, const host::Fr& alpha, const host::Fr& alpha_base, hipStream_t st)
{
    const host::Fr f2 = host::fr_from_u64(32), f3 = host::fr_from_u64(1024);
    A.alpha_m256 = host::limbs_m256(alpha);
    A.alpha_fix_m261 = host::limbs_m261(host::fr_mul(alpha, f2));
    A.alpha2_fix_m261 = host::limbs_m261(host::fr_mul(host::fr_sqr(alpha), f2));
    A.abase_m261 = host::limbs_m261(alpha_base);
    A.abase_fix2_m261 = host::limbs_m261(host::fr_mul(alpha_base, f2));
    A.abase_fix3_m261 = host::limbs_m261(host::fr_mul(alpha_base, f3));
    k_quotient_mid<<<pw_blocks(A.n2), PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int quotient_mimc(QuotMimcArgs A, const host::Fr& alpha_base, const host::Fr& alpha_step, hipStream_t st)
{
    A.alpha_m261 = host::limbs_m261(alpha_step);
    A.abase_fix_m261 = host::limbs_m261(host::fr_mul(alpha_base, host::fr_from_u64(32)));
    k_quotient_mimc<<<pw_blocks(A.n4), PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int quotient_seq(QuotSeqArgs A, const host::Fr& c, hipStream_t st)
{
    A.c_fix_m261 = host::limbs_m261(host::fr_mul(c, host::fr_from_u64(32)));
    k_quotient_seq<<<pw_blocks(A.n2), PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int quotient_bool(QuotBoolArgs A, const host::Fr& c_left, const host::Fr& c_right, const host::Fr& c_out, hipStream_t st)
{
    const host::Fr f2 = host::fr_from_u64(32);
    A.cl_fix_m261 = host::limbs_m261(host::fr_mul(c_left, f2));
    A.cr_fix_m261 = host::limbs_m261(host::fr_mul(c_right, f2));
    A.co_fix_m261 = host::limbs_m261(host::fr_mul(c_out, f2));
    k_quotient_bool<<<pw_blocks(A.n2), PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

// g^n for n = 2^log2n
static host::Fr coset_gen_pow_n(int log2n)
{
    host::Fr a = host::fr_from_limbs(FrHostP::GEN5);
    for (int i = 0; i < log2n; i++) a = host::fr_sqr(a);
    return a;
}

// divide_by_pseudo_vanishing_polynomial(coeffs, src = 2^log2n, target = 2^log2N), in place on the resident coset evaluations
int divide_by_pseudo_vanishing(uint64_t* d_coeffs, int log2n, int log2N, hipStream_t st)
{
    const int lk = log2N - log2n;
    if (lk < 0 || lk > 2) {
        set_error("divide_by_pseudo_vanishing: target / source domain ratio must be 1, 2 or 4");
        return BBGPU_ERR_SIZE;
    }
    const uint32_t k = 1u << lk;
    const size_t N = (size_t)1 << log2N;
    // (g w_N^i)^n - 1 = g^n w_k^(i mod k) - 1  (compute_multiplicative_subgroup, :104-127)
    host::Fr sub[4], acc = coset_gen_pow_n(log2n), wk = host::fr_root_of_unity(lk);
    Limbs9 inv[4];
    for (uint32_t j = 0; j < 4; j++) {
        if (j < k) {
            sub[j] = host::fr_inv(host::fr_sub(acc, host::fr_one()));
            acc = host::fr_mul(acc, wk);
        } else {
            sub[j] = host::fr_zero();
        }
        inv[j] = host::limbs_m261(sub[j]);
    }
    const host::Fr rootN = host::fr_root_of_unity(log2N), wninv = host::fr_inv(host::fr_root_of_unity(log2n));
    const uint32_t blocks = strided_blocks(N);
    k_divide_vanishing<<<blocks, PT, 0, st>>>((uint32_t*)d_coeffs, (uint32_t)N, k, make_powtab(rootN), host::limbs_m261(host::fr_from_limbs(FrHostP::GEN5)),
                                             host::limbs_m261(host::fr_pow(rootN, (uint64_t)blocks * PT)), host::limbs_m261(wninv), inv[0], inv[1], inv[2], inv[3]);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

// compute_lagrange_polynomial_fft(l_1, src = 2^log2n, target = 2^log2N): N resident values; d_tmp: N elements of workspace
int lagrange_l1_fft(uint64_t* d_l1, uint64_t* d_tmp, int log2n, int log2N, Scratch& S, hipStream_t st)
{
    const int lk = log2N - log2n;
    if (lk < 0 || lk > 2) {
        set_error("lagrange_l1_fft: target / source domain ratio must be 1, 2 or 4");
        return BBGPU_ERR_SIZE;
    }
    const uint32_t k = 1u << lk;
    const size_t N = (size_t)1 << log2N;
    const host::Fr rootN = host::fr_root_of_unity(log2N);
    const uint32_t blocks = strided_blocks(N);
    k_l1_denominators<<<blocks, PT, 0, st>>>((uint32_t*)d_l1, (uint32_t)N, make_powtab(rootN), host::limbs_m256(host::fr_from_limbs(FrHostP::GEN5)),
                                            host::limbs_m261(host::fr_pow(rootN, (uint64_t)blocks * PT)));
    HIPCHK(launch_check());
    int rc = batch_invert(d_l1, d_tmp, N, S, st);
    if (rc) return rc;
    // numerators ((g w)^n - 1) / n: k values
    host::Fr acc = coset_gen_pow_n(log2n), wk = host::fr_root_of_unity(lk);
    const host::Fr ninv = host::fr_inv(host::fr_from_u64((uint64_t)1 << log2n));
    Limbs9 s[4];
    for (uint32_t j = 0; j < 4; j++) {
        host::Fr v = host::fr_zero();
        if (j < k) {
            v = host::fr_mul(host::fr_sub(acc, host::fr_one()), ninv);
            acc = host::fr_mul(acc, wk);
        }
        s[j] = host::limbs_m261(v);
    }
    k_l1_scale<<<pw_blocks(N), PT, 0, st>>>((uint32_t*)d_l1, (uint32_t)N, k, s[0], s[1], s[2], s[3]);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

int lincomb(LinCombArgs A, const host::Fr* coeffs, hipStream_t st)
{
    for (int j = 0; j < A.count; j++) A.c[j] = host::limbs_m261(coeffs[j]);
    k_lincomb<<<pw_blocks(A.n), PT, 0, st>>>(A);
    HIPCHK(launch_check());
    return BBGPU_OK;
}

} // namespace poly
} // namespace bbgpu
