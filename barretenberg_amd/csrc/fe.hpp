// fe.hpp -- 256-bit prime-field arithmetic for gfx950 as integer MAC chains.
//
// Replaces, on the device, the reference's 4x64-bit Montgomery layer
// (src/barretenberg/fields/field_impl_int128.tcc:72-137,149-263; asm twin asm_macros.hpp:141-413).
//
// Representation chosen from measurements on MI355X (tools/ubench): v_mad_u64_u32 issues at ~half the rate of a
// plain 32-bit add and at the SAME rate as an add-with-carry, so carry chains are as expensive as multiplies.
// We therefore use 9 limbs of 29 bits with lazy carries: a 9x9 product-scanning multiply accumulates up to 18
// partial products per 64-bit column with no carry handling at all (163 v_mad_u64_u32 + ~50 cheap ops, vs 132 mads
// + ~570 carry/move ops for the compiler's 4x64 __int128 code).  Montgomery radix is R = 2^261.
//
// Lazy bounds are tracked in the TYPE so that overflow is a compile error, not a silent GPU bug:
//   Fe<F, L, V>:  every limb d[i] < L * U  (U = 2^29 + 8),  integer value < V * p,  value < 2^261.
// add/sub never reduce; mul/sqr accept any operands with L1*L2 <= 6 (normalising automatically otherwise) and
// return L = 1 with V = V1*V2/169 + 2 (since 2^261 / p > 169).  No conditional subtraction exists anywhere except in
// to_canonical().
//
// Everything is __host__ __device__ so the same code is unit-tested on the CPU against the oracle.
#pragma once
#include <stdint.h>

#include "bn254_params.h"

#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__) && !defined(BBGPU_NO_MONT_ASM)
#define BBGPU_MONT_ASM 1
#include "fe_mont_gfx950.h"
#else
#define BBGPU_MONT_ASM 0
#endif

#if defined(__HIPCC__)
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline __attribute__((always_inline))
#endif

namespace bbgpu {

constexpr uint32_t M29 = 0x1fffffffu;
constexpr int NL = 9;
constexpr int MAXL = 7;   // 7 * (2^29 + 8) < 2^32
constexpr int MAXV = 168; // 168 * p < 2^261

template <class F, int L, int V> struct Fe {
    static_assert(L >= 1 && L <= MAXL, "limb bound overflows 32 bits: insert weak()");
    static_assert(V >= 1 && V <= MAXV, "value bound overflows 2^261: insert reduce_value()");
    uint32_t d[NL];
    BB_HD Fe() {}
    template <int L2, int V2> BB_HD Fe(const Fe<F, L2, V2>& o)
    {
        static_assert(L2 <= L && V2 <= V, "narrowing a lazy bound");
#pragma unroll
        for (int i = 0; i < NL; i++) d[i] = o.d[i];
    }
};

template <class F> using FeT = Fe<F, 1, 1>;   // canonical constant
template <class F> using FeN = Fe<F, 1, 12>;  // the storage type of kernels: tight limbs, value < 12p

// The same with EXACT limbs: d[0..7] < 2^29 (what carry_full delivers; L = 1 only promises < 2^29 + 8).  Subtracting such a value
// needs a borrow-proofing offset of one 2^29 instead of two, which keeps the difference one unit tighter: sub(a, FeE) has
// L = L1 + 2.  That unit decides whether Y3 = R (Q - X3) - Y1 PPP of the mixed addition fits one shared reduction without
// renormalising -Y1 first (g1.hpp madd_ip).
template <class F, int V> struct FeE : Fe<F, 1, V> {
    BB_HD FeE() {}
};
// mul / sqr / mul_add results and unpack() have exact limbs by construction (every limb but the top one is masked to 29 bits, the top one is
// below 2^29 because the value is below 2^261): say so where a subtraction can use it.  For those sources ONLY.
template <class F, int V> BB_HD FeE<F, V> exact_limbs(const Fe<F, 1, V>& product_or_unpacked)
{
    FeE<F, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = product_or_unpacked.d[i];
    return r;
}

// ---- constants --------------------------------------------------------------------------------------------------
struct Limbs9 {
    uint32_t d[NL];
};
// K * p written so that every limb is >= Lb * U (borrow-proofed), for computing a - b as a + (Kp - b)
template <class F> constexpr Limbs9 make_sub_const(int Lb, int K)
{
    Limbs9 c{};
    uint64_t carry = 0;
    for (int i = 0; i < NL; i++) {
        uint64_t t = (uint64_t)F::P[i] * (uint64_t)K + carry;
        c.d[i] = (i == NL - 1) ? (uint32_t)t : (uint32_t)(t & M29);
        carry = t >> 29;
    }
    const uint32_t off = (uint32_t)(Lb + 1) << 29; // (Lb+1) * 2^29 added to limb i, (Lb+1) taken from limb i+1
    for (int i = 0; i < NL; i++) {
        if (i < NL - 1) c.d[i] += off;
        if (i > 0) c.d[i] -= (uint32_t)(Lb + 1);
    }
    return c;
}

template <class F> BB_HD FeT<F> fe_from(const uint32_t (&a)[NL])
{
    FeT<F> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a[i];
    return r;
}
template <class F> BB_HD FeT<F> fe_zero()
{
    FeT<F> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = 0;
    return r;
}
template <class F> BB_HD FeT<F> fe_one()
{
    return fe_from<F>(F::ONE);
}

// ---- add / sub (no reduction, no carries) -----------------------------------------------------------------------
template <class F, int L1, int V1, int L2, int V2>
BB_HD Fe<F, L1 + L2, V1 + V2> add(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b)
{
    Fe<F, L1 + L2, V1 + V2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <class F, int L1, int V1> BB_HD Fe<F, 2 * L1, 2 * V1> dbl(const Fe<F, L1, V1>& a)
{
    return add(a, a);
}
// a - b + (V2+1)p, limb-wise non-negative
template <class F, int L1, int V1, int L2, int V2>
BB_HD Fe<F, L1 + L2 + 2, V1 + V2 + 1> sub(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b)
{
    constexpr Limbs9 c = make_sub_const<F>(L2, V2 + 1);
    Fe<F, L1 + L2 + 2, V1 + V2 + 1> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i] + c.d[i] - b.d[i];
    return r;
}
// the same for a subtrahend with exact limbs: offset 2^29 per limb, result < L1*U + 2*2^29
template <class F, int L1, int V1, int V2>
BB_HD Fe<F, L1 + 2, V1 + V2 + 1> sub(const Fe<F, L1, V1>& a, const FeE<F, V2>& b)
{
    constexpr Limbs9 c = make_sub_const<F>(0, V2 + 1);
    Fe<F, L1 + 2, V1 + V2 + 1> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i] + c.d[i] - b.d[i];
    return r;
}
// -a + (V+1)p
template <class F, int L1, int V1> BB_HD Fe<F, L1 + 2, V1 + 1> neg(const Fe<F, L1, V1>& a)
{
    constexpr Limbs9 c = make_sub_const<F>(L1, V1 + 1);
    Fe<F, L1 + 2, V1 + 1> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = c.d[i] - a.d[i];
    return r;
}

// one parallel carry round: limbs back below 2^29 + 8, value unchanged
template <class F, int L, int V> BB_HD Fe<F, 1, V> weak(const Fe<F, L, V>& a)
{
    Fe<F, 1, V> r;
    r.d[0] = a.d[0] & M29;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.d[i] = (a.d[i] & M29) + (a.d[i - 1] >> 29);
    r.d[NL - 1] = a.d[NL - 1] + (a.d[NL - 2] >> 29);
    return r;
}
template <class F, int V> BB_HD Fe<F, 1, V> weak(const Fe<F, 1, V>& a)
{
    return a;
}

// full sequential carry: limbs 0..7 exactly < 2^29 (unique representation of the integer value)
template <class F, int L, int V> BB_HD FeE<F, V> carry_full(const Fe<F, L, V>& a)
{
    FeE<F, V> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        uint32_t t = a.d[i] + c;
        r.d[i] = t & M29;
        c = t >> 29;
    }
    r.d[NL - 1] = a.d[NL - 1] + c;
    return r;
}

// ---- Montgomery multiplication, R = 2^261 -----------------------------------------------------------------------
// every column sum is <= 9*(L1*U)*(L2*U) + 9*2^58 + carry < 2^64 whenever L1*L2 <= 6.
// On the device the three products below are the hand-scheduled gfx950 sequences of fe_mont_gfx950.h (generated by
// tools/gen_mont_asm.py): hipcc schedules the C++ form with a separate 64-bit addition per column (v_lshl_add_u64) and
// 277 instructions per multiplication, the asm form takes the carry as the addend of the column's first v_mad_u64_u32 (205).
// The C++ form stays the definition: it is what the host build runs (tests/cpp/test_fe_host.cpp) and what
// -DBBGPU_NO_MONT_ASM selects on the device (A/B, and the device self-test compares both against the reference's vectors).
BB_HD uint64_t mad_carry_in(uint32_t a, uint32_t b, uint64_t c)
{
    return (uint64_t)a * b + c;
}

// Fused product scanning: one 64-bit accumulator walks the 18 columns; the carry out of column k is the addend of the
// first multiply-add of column k+1, so there is no column array and no 64-bit carry addition.
template <class F> BB_HD void mul_raw(const uint32_t (&a)[NL], const uint32_t (&b)[NL], uint32_t (&out)[NL])
{
#if BBGPU_MONT_ASM
    mul_raw_gfx950<F>(a, b, out);
    return;
#endif
    uint32_t m[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc = mad_carry_in(a[0], b[k], acc);
#pragma unroll
        for (int i = 1; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::P[k - i];
        m[k] = ((uint32_t)acc * F::PINV) & M29;
        acc += (uint64_t)m[k] * F::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
        acc = mad_carry_in(a[k - (NL - 1)], b[NL - 1], acc);
#pragma unroll
        for (int i = k - (NL - 1) + 1; i < NL; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * F::P[k - i];
        out[k - NL] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    out[NL - 1] = (uint32_t)acc;
}

template <class F> BB_HD void sqr_raw(const uint32_t (&a)[NL], uint32_t (&out)[NL])
{
#if BBGPU_MONT_ASM
    sqr_raw_gfx950<F>(a, out);
    return;
#endif
    uint32_t a2[NL], m[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; k++) {
        // first product of the column takes the carry as its addend: the square term on even columns, else the first cross term
        if ((k & 1) == 0) {
            acc = mad_carry_in(a[k / 2], a[k / 2], acc);
        } else {
            const int i0 = k < NL ? 0 : k - (NL - 1);
            acc = mad_carry_in(a2[i0], a[k - i0], acc);
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int j = k - i;
            const int i0 = k < NL ? 0 : k - (NL - 1);
            if (j > i && j < NL && !((k & 1) == 1 && i == i0)) acc += (uint64_t)a2[i] * a[j];
        }
        if (k < NL) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::P[k - i];
            m[k] = ((uint32_t)acc * F::PINV) & M29;
            acc += (uint64_t)m[k] * F::P[0];
        } else {
#pragma unroll
            for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * F::P[k - i];
            out[k - NL] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    out[NL - 1] = (uint32_t)acc;
}

// a*b + c*d with ONE Montgomery reduction (81 multiply-adds saved): columns hold 18 + 18 products, so the limb bounds
// must satisfy L1*L2 + L3*L4 <= 6.
template <class F>
BB_HD void mul2_raw(const uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&c)[NL], const uint32_t (&d)[NL], uint32_t (&out)[NL])
{
#if BBGPU_MONT_ASM
    mul2_raw_gfx950<F>(a, b, c, d, out);
    return;
#endif
    uint32_t m[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc = mad_carry_in(a[0], b[k], acc);
        acc += (uint64_t)c[0] * d[k];
#pragma unroll
        for (int i = 1; i <= k; i++) {
            acc += (uint64_t)a[i] * b[k - i];
            acc += (uint64_t)c[i] * d[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::P[k - i];
        m[k] = ((uint32_t)acc * F::PINV) & M29;
        acc += (uint64_t)m[k] * F::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
        acc = mad_carry_in(a[k - (NL - 1)], b[NL - 1], acc);
        acc += (uint64_t)c[k - (NL - 1)] * d[NL - 1];
#pragma unroll
        for (int i = k - (NL - 1) + 1; i < NL; i++) {
            acc += (uint64_t)a[i] * b[k - i];
            acc += (uint64_t)c[i] * d[k - i];
        }
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * F::P[k - i];
        out[k - NL] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    out[NL - 1] = (uint32_t)acc;
}

// In-place forms: a <- a*b, c <- a*b + c*d.  On the device the result takes the REGISTERS of the replaced operand (fe_mont_gfx950.h:
// operand limb j is last read in column j + 8, result limb j is written after column j + 9), so a loop-carried value that is
// multiplied in place needs no copy on the loop's back edge; elsewhere they are the plain products.
template <class F> BB_HD void mul_raw_inplace(uint32_t (&a)[NL], const uint32_t (&b)[NL])
{
#if BBGPU_MONT_ASM
    mul_raw_inplace_gfx950<F>(a, b);
    return;
#endif
    uint32_t t[NL];
    mul_raw<F>(a, b, t);
#pragma unroll
    for (int i = 0; i < NL; i++) a[i] = t[i];
}
template <class F>
BB_HD void mul2_raw_inplace(const uint32_t (&a)[NL], const uint32_t (&b)[NL], uint32_t (&c)[NL], const uint32_t (&d)[NL])
{
#if BBGPU_MONT_ASM
    mul2_raw_inplace_gfx950<F>(a, b, c, d);
    return;
#endif
    uint32_t t[NL];
    mul2_raw<F>(a, b, c, d, t);
#pragma unroll
    for (int i = 0; i < NL; i++) c[i] = t[i];
}

// "addhi" forms (round 4): a <- REDC(a b) + e and out = REDC(a^2) + e, computed as REDC(a b + e 2^261) -- limb j of e is added to column j + 9 of the
// double-width product, inside the carry chain the reduction runs anyway.  The sum comes out with EXACT limbs, so a difference such as
// P = x2 ZZ1 - X1 of the mixed addition (e = K p - X1, limb-wise non-negative) costs nine additions of e's limbs instead of a separate
// subtraction (18 instructions) and renormalisation (24), and X3 = R^2 - (PPP + 2 Q) needs no carry_full (27).  e may have any limbs below 2^32:
// a column gains at most 2^32 against a head-room of 2^58 (9 (6 U^2) + 9 2^58 + carry < 2^64).
template <class F> BB_HD void mul_addhi_raw_inplace(uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&e)[NL])
{
#if BBGPU_MONT_ASM
    mul_addhi_raw_inplace_gfx950<F>(a, b, e);
    return;
#endif
    uint32_t m[NL], out[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        acc = mad_carry_in(a[0], b[k], acc);
#pragma unroll
        for (int i = 1; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * F::P[k - i];
        m[k] = ((uint32_t)acc * F::PINV) & M29;
        acc += (uint64_t)m[k] * F::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
        acc = mad_carry_in(a[k - (NL - 1)], b[NL - 1], acc);
#pragma unroll
        for (int i = k - (NL - 1) + 1; i < NL; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - (NL - 1); i < NL; i++) acc += (uint64_t)m[i] * F::P[k - i];
        acc += e[k - NL];
        out[k - NL] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    out[NL - 1] = (uint32_t)acc + e[NL - 1];
#pragma unroll
    for (int i = 0; i < NL; i++) a[i] = out[i];
}
template <class F> BB_HD void sqr_addhi_raw(const uint32_t (&a)[NL], const uint32_t (&e)[NL], uint32_t (&out)[NL])
{
#if BBGPU_MONT_ASM
    sqr_addhi_raw_gfx950<F>(a, e, out);
    return;
#endif
    uint32_t t[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) t[i] = a[i];
    mul_addhi_raw_inplace<F>(t, a, e); // the host form: the plain product (the device form saves the 36 symmetric multiply-adds)
#pragma unroll
    for (int i = 0; i < NL; i++) out[i] = t[i];
}

constexpr int mul_v(int v1, int v2)
{
    return (v1 * v2) / 169 + 2;
}

template <class F, int L1, int V1, int L2, int V2>
BB_HD Fe<F, 1, mul_v(V1, V2)> mul(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b)
{
    static_assert(mul_v(V1, V2) <= MAXV, "product value bound too large");
    Fe<F, 1, mul_v(V1, V2)> r;
    if constexpr (L1 * L2 <= 6) {
        mul_raw<F>(a.d, b.d, r.d);
    } else if constexpr (L1 >= L2 && L2 <= 6) {
        Fe<F, 1, V1> an = weak(a);
        mul_raw<F>(an.d, b.d, r.d);
    } else if constexpr (L2 > L1 && L1 <= 6) {
        Fe<F, 1, V2> bn = weak(b);
        mul_raw<F>(a.d, bn.d, r.d);
    } else {
        Fe<F, 1, V1> an = weak(a);
        Fe<F, 1, V2> bn = weak(b);
        mul_raw<F>(an.d, bn.d, r.d);
    }
    return r;
}

constexpr int mul2_v(int v1, int v2, int v3, int v4)
{
    return (v1 * v2 + v3 * v4) / 169 + 2;
}
// a*b + c*d, one reduction
template <class F, int L1, int V1, int L2, int V2, int L3, int V3, int L4, int V4>
BB_HD Fe<F, 1, mul2_v(V1, V2, V3, V4)> mul_add(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b, const Fe<F, L3, V3>& c, const Fe<F, L4, V4>& d)
{
    static_assert(L1 * L2 + L3 * L4 <= 6, "limb bounds too large for a shared reduction: weak() an operand");
    static_assert(mul2_v(V1, V2, V3, V4) <= MAXV, "value bound too large");
    Fe<F, 1, mul2_v(V1, V2, V3, V4)> r;
    mul2_raw<F>(a.d, b.d, c.d, d.d, r.d);
    return r;
}
// a*b - c*d = a*b + (Kp - c)*d, one reduction
template <class F, int L1, int V1, int L2, int V2, int L3, int V3, int L4, int V4>
BB_HD auto mul_sub(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b, const Fe<F, L3, V3>& c, const Fe<F, L4, V4>& d)
{
    return mul_add(a, b, weak(neg(c)), d);
}

// a*b with the result in a's registers (use where a dies here); no automatic renormalisation: the caller's bounds must fit
template <class F, int L1, int V1, int L2, int V2>
BB_HD Fe<F, 1, mul_v(V1, V2)> mul_ip(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b)
{
    static_assert(L1 * L2 <= 6, "limb bounds too large: weak() an operand");
    static_assert(mul_v(V1, V2) <= MAXV, "product value bound too large");
    Fe<F, 1, mul_v(V1, V2)> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i];
    mul_raw_inplace<F>(r.d, b.d);
    return r;
}
// a*b + c*d with the result in c's registers (one reduction)
template <class F, int L1, int V1, int L2, int V2, int L3, int V3, int L4, int V4>
BB_HD Fe<F, 1, mul2_v(V1, V2, V3, V4)> mul_add_ip(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b, const Fe<F, L3, V3>& c, const Fe<F, L4, V4>& d)
{
    static_assert(L1 * L2 + L3 * L4 <= 6, "limb bounds too large for a shared reduction: weak() an operand");
    static_assert(mul2_v(V1, V2, V3, V4) <= MAXV, "value bound too large");
    Fe<F, 1, mul2_v(V1, V2, V3, V4)> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = c.d[i];
    mul2_raw_inplace<F>(a.d, b.d, r.d, d.d);
    return r;
}

// REDC(a b) + e with the result in a's registers and exact limbs (see mul_addhi_raw_inplace); e: any limb bound
template <class F, int L1, int V1, int L2, int V2, int L3, int V3>
BB_HD FeE<F, mul_v(V1, V2) + V3> mul_addhi_ip(const Fe<F, L1, V1>& a, const Fe<F, L2, V2>& b, const Fe<F, L3, V3>& e)
{
    static_assert(L1 * L2 <= 6, "limb bounds too large: weak() an operand");
    static_assert(mul_v(V1, V2) + V3 <= MAXV, "value bound too large");
    FeE<F, mul_v(V1, V2) + V3> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = a.d[i];
    mul_addhi_raw_inplace<F>(r.d, b.d, e.d);
    return r;
}
// REDC(a^2) + e, exact limbs
template <class F, int L1, int V1, int L3, int V3> BB_HD FeE<F, mul_v(V1, V1) + V3> sqr_addhi(const Fe<F, L1, V1>& a, const Fe<F, L3, V3>& e)
{
    static_assert(L1 * L1 <= 6, "limb bound too large for the squaring: weak() the operand");
    static_assert(mul_v(V1, V1) + V3 <= MAXV, "value bound too large");
    FeE<F, mul_v(V1, V1) + V3> r;
    sqr_addhi_raw<F>(a.d, e.d, r.d);
    return r;
}

template <class F, int L1, int V1> BB_HD Fe<F, 1, mul_v(V1, V1)> sqr(const Fe<F, L1, V1>& a)
{
    Fe<F, 1, mul_v(V1, V1)> r;
    if constexpr (L1 * L1 <= 6) {
        sqr_raw<F>(a.d, r.d);
    } else {
        Fe<F, 1, V1> an = weak(a);
        sqr_raw<F>(an.d, r.d);
    }
    return r;
}

// value < V*p  ->  value' == value (mod p), value' < 3p, tight limbs.  ~60 instructions.
// (reduce_value_raw: the same without the final carry round, limbs up to 3 U -- for a consumer that carries anyway, to_canonical)
template <class F, int L, int V> BB_HD Fe<F, 3, 3> reduce_value_raw(const Fe<F, L, V>& a)
{
    Fe<F, 1, V> t = carry_full(a);
    // q <= floor(value / p), q >= that - 2
    const uint32_t q = (uint32_t)(((uint64_t)t.d[NL - 1] * F::QMAGIC) >> 32);
    // qp = q * p in exact 29-bit limbs
    uint32_t qp[NL];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint64_t m = (uint64_t)q * F::P[i] + c;
        qp[i] = (i == NL - 1) ? (uint32_t)m : ((uint32_t)m & M29);
        c = m >> 29;
    }
    // t - qp >= 0: borrow-proof by lending 2^29 to every limb below the top
    Fe<F, 3, 3> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t v = t.d[i] - qp[i];
        if (i < NL - 1) v += (1u << 29);
        if (i > 0) v -= 1u;
        r.d[i] = v;
    }
    return r;
}
template <class F, int L, int V> BB_HD Fe<F, 1, 3> reduce_value(const Fe<F, L, V>& a)
{
    return weak(reduce_value_raw(a));
}

// ---- memory format: 8 x u32 little-endian words (== the reference's 4 x u64 field_t) ---------------------------
template <class F> BB_HD Fe<F, 1, 6> unpack(const uint32_t (&w)[8])
{
    Fe<F, 1, 6> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 29 * i, j = bit >> 5, s = bit & 31;
        uint32_t lo = w[j] >> s;
        if (s > 3 && j + 1 < 8) lo |= w[j + 1] << (32 - s);
        r.d[i] = lo & M29;
    }
    return r;
}
// limbs must be exact (carry_full) and value < 2^256
template <class F, int V> BB_HD void pack_exact(const Fe<F, 1, V>& a, uint32_t (&w)[8])
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // word j covers bits [32j, 32j+32): limbs floor(32j/29) .. floor((32j+31)/29)
        const int lo_l = (32 * j) / 29, hi_l = (32 * j + 31) / 29;
        uint32_t v = 0;
#pragma unroll
        for (int l = lo_l; l <= hi_l; l++) {
            if (l < NL) {
                const int sh = 29 * l - 32 * j;
                v |= (sh >= 0) ? (a.d[l] << sh) : (a.d[l] >> (-sh));
            }
        }
        w[j] = v;
    }
}
template <class F, int L, int V> BB_HD void pack(const Fe<F, L, V>& a, uint32_t (&w)[8])
{
    static_assert(V <= 5, "value may not fit 256 bits: reduce_value() first");
    pack_exact(carry_full(a), w);
}

// conditional subtraction of a 256-bit constant held as 4 x u64
BB_HD void cond_sub_256(uint32_t (&w)[8], const uint64_t (&m)[4])
{
    uint32_t t[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t mi = (uint32_t)(m[i >> 1] >> ((i & 1) * 32));
        uint64_t dif = (uint64_t)w[i] - mi - borrow;
        t[i] = (uint32_t)dif;
        borrow = (dif >> 63) & 1;
    }
    const bool keep = borrow != 0; // w < m
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = keep ? w[i] : t[i];
}

// any lazy value -> canonical [0, p) as 8 words
template <class F, int L, int V> BB_HD void to_canonical(const Fe<F, L, V>& a, uint32_t (&w)[8])
{
    if constexpr (V <= 4) {
        pack_exact(carry_full(a), w);
        if constexpr (V > 2) cond_sub_256(w, F::P2_64);
        if constexpr (V > 1) cond_sub_256(w, F::P64);
    } else {
        pack_exact(carry_full(reduce_value_raw(a)), w);
        cond_sub_256(w, F::P2_64);
        cond_sub_256(w, F::P64);
    }
}

// the same for a value whose limbs are already exact (a product): no carry chain
template <class F, int V> BB_HD void to_canonical(const FeE<F, V>& a, uint32_t (&w)[8])
{
    static_assert(V <= 4, "value may not fit 256 bits");
    pack_exact(a, w);
    if constexpr (V > 2) cond_sub_256(w, F::P2_64);
    if constexpr (V > 1) cond_sub_256(w, F::P64);
}

// k * p in exact 29-bit limbs
template <class F> constexpr Limbs9 make_multiple(int K)
{
    Limbs9 c{};
    uint64_t carry = 0;
    for (int i = 0; i < NL; i++) {
        uint64_t t = (uint64_t)F::P[i] * (uint64_t)K + carry;
        c.d[i] = (i == NL - 1) ? (uint32_t)t : (uint32_t)(t & M29);
        carry = t >> 29;
    }
    return c;
}
// exact-limb zero test on a multiplication result (limbs exact, value < V*p, V <= 3):
// v == 0 (mod p)  <=>  v in {0, p, 2p}
template <class F, int V> BB_HD bool is_zero_mulout(const Fe<F, 1, V>& a)
{
    static_assert(V <= 3, "only valid on a fresh product");
    constexpr Limbs9 p2 = make_multiple<F>(2);
    uint32_t z = 0, e = 0, e2 = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        z |= a.d[i];
        e |= a.d[i] ^ F::P[i];
        e2 |= a.d[i] ^ p2.d[i];
    }
    return z == 0 || e == 0 || (V > 2 && e2 == 0);
}
// generic (slow) zero test mod p
template <class F, int L, int V> BB_HD bool is_zero_slow(const Fe<F, L, V>& a)
{
    uint32_t w[8];
    to_canonical(a, w);
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) z |= w[i];
    return z == 0;
}
template <class F, int L, int V> BB_HD bool limbs_all_zero(const Fe<F, L, V>& a)
{
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) z |= a.d[i];
    return z == 0;
}

// ---- conversions between the reference's Montgomery form (x * 2^256) and ours (x * 2^261) ----------------------
template <class F> BB_HD Fe<F, 1, 2> m256_to_m261(const Fe<F, 1, 6>& a)
{
    return mul(a, fe_from<F>(F::M256_TO_M261));
}
template <class F, int L, int V> BB_HD Fe<F, 1, 2> m261_to_m256(const Fe<F, L, V>& a)
{
    static_assert(mul_v(V, 1) <= 2, "");
    return mul(a, fe_from<F>(F::M261_TO_M256));
}

// a^e for a 256-bit exponent given as 4 x u64 (used for the single inversion at the end of an MSM and for tests)
template <class F> BB_HD Fe<F, 1, 2> pow_u256(const Fe<F, 1, 2>& a, const uint64_t (&e)[4])
{
    Fe<F, 1, 2> acc = fe_one<F>();
    for (int i = 255; i >= 0; --i) {
        acc = sqr(acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a);
    }
    return acc;
}

} // namespace bbgpu
