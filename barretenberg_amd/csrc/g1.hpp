// g1.hpp -- BN254 G1 (y^2 = x^3 + 3) group law for the MSM kernels, extended-Jacobian ("XYZZ") coordinates:
//   x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2; infinity <=> ZZ == 0 (all limbs zero).
// Replaces, on the device, g1::mixed_add / add / dbl (src/barretenberg/groups/group.hpp:153-448).  The reference
// uses Jacobian (7M+4S mixed add, 11M+5S add); XYZZ needs 8M+2S / 12M+2S with cheaper squarings and, more
// importantly on a GPU, no dependent Z^2,Z^3 recomputation.  Any representative is fine: results only leave the
// device after normalisation, which is unique.
// All exceptional cases of the group law (P+P, P+(-P), infinity operands) are handled exactly, because outputs are
// compared bit-for-bit with the reference after normalisation.
#pragma once
#include "fe.hpp"

namespace bbgpu {

using Fq = FqP;
using FqN = FeN<Fq>;

// device working form: Montgomery-261, tight limbs, value < V*p; never infinity (reference: group.hpp:311-312)
template <int V> struct AffineV {
    Fe<Fq, 1, V> x, y;
};
using Affine = AffineV<4>;

struct Xyzz {
    FqN x, y, zz, zzz;
};

BB_HD void set_infinity(Xyzz& p)
{
    FeT<Fq> z = fe_zero<Fq>();
    p.x = z;
    p.y = z;
    p.zz = z;
    p.zzz = z;
}
BB_HD bool is_infinity(const Xyzz& p)
{
    return limbs_all_zero(p.zz);
}
template <int V> BB_HD void from_affine(Xyzz& r, const AffineV<V>& a)
{
    r.x = a.x;
    r.y = a.y;
    r.zz = fe_one<Fq>();
    r.zzz = fe_one<Fq>();
}

// 2 * (affine) -> XYZZ   [mdbl-2008-s-1]; y != 0 on this curve (no 2-torsion)
template <int V> BB_HD void dbl_affine(Xyzz& r, const AffineV<V>& a)
{
    auto U = weak(dbl(a.y));                       // 2Y
    auto Vv = sqr(U);                              // V = U^2
    auto W = mul(U, Vv);                           // W = U*V
    auto S = mul(a.x, Vv);                         // S = X*V
    auto XX = sqr(a.x);
    auto M = weak(add(dbl(XX), XX));               // 3*X^2 (a = 0)
    auto X3 = weak(sub(sqr(M), dbl(S)));           // M^2 - 2S
    auto Y3 = mul_sub(M, sub(S, X3), W, a.y);
    r.x = X3;
    r.y = Y3;
    r.zz = Vv;
    r.zzz = W;
}

// 2 * XYZZ  [dbl-2008-s-1]
BB_HD void dbl(Xyzz& r, const Xyzz& p)
{
    if (is_infinity(p)) {
        set_infinity(r);
        return;
    }
    auto U = weak(dbl(p.y));
    auto Vv = sqr(U);
    auto W = mul(U, Vv);
    auto S = mul(p.x, Vv);
    auto XX = sqr(p.x);
    auto M = weak(add(dbl(XX), XX));
    auto X3 = weak(sub(sqr(M), dbl(S)));
    auto Y3 = mul_sub(M, sub(S, X3), W, p.y);
    auto ZZ3 = mul(Vv, p.zz);
    auto ZZZ3 = mul(W, p.zzz);
    r.x = X3;
    r.y = Y3;
    r.zz = ZZ3;
    r.zzz = ZZZ3;
}

// acc += a   [madd-2008-s], 8M + 2S.  Exceptional cases: acc == inf, acc == a, acc == -a.
template <int V> BB_HD void madd(Xyzz& acc, const AffineV<V>& a)
{
    if (is_infinity(acc)) {
        from_affine(acc, a);
        return;
    }
    auto U2 = mul(a.x, acc.zz);
    auto S2 = mul(a.y, acc.zzz);
    auto P = weak(sub(U2, acc.x));
    auto R = weak(sub(S2, acc.y));
    auto PP = sqr(P);
    if (is_zero_mulout(PP)) {  // same x: rare
        if (is_zero_slow(R)) {
            dbl_affine(acc, a);
        } else {
            set_infinity(acc);
        }
        return;
    }
    auto PPP = mul(P, PP);
    auto Q = mul(acc.x, PP);
    auto X3 = weak(sub(sqr(R), add(PPP, dbl(Q))));
    auto Y3 = mul_sub(R, sub(Q, X3), acc.y, PPP);
    auto ZZ3 = mul(acc.zz, PP);
    auto ZZZ3 = mul(acc.zzz, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = ZZ3;
    acc.zzz = ZZZ3;
}

// The same for the bucket accumulation's hot loop (msm_accumulate_kernel), written around what that loop pays for besides its ten
// products (round 3; the instruction counts are those of the gfx950 ISA, DESIGN_HISTORY.md 5):
//  * the accumulator's infinity is a FLAG (set at a bucket start and by P + (-P)); the caller takes the `acc = a` branch itself, so
//    this function only sees a finite accumulator -- and sets the flag (and a clean infinity) when the sum cancels;
//  * the results land in the accumulator's own registers: ZZ3, ZZZ3 and Y3 are in-place products (mul_ip / mul_add_ip, the result
//    takes the registers of the operand it replaces), X3 is plain limb arithmetic the register allocator coalesces -- so the loop's
//    back edge carries no copies (round 2: 36 v_mov per addition);
//  * X3 is carried out EXACTLY (carry_full, 27 dependent instructions instead of weak()'s 24): with an exact subtrahend Q - X3 has
//    L = 3 and R (Q - X3) + (-Y1) PPP fits the shared reduction with -Y1 left at L = 3, i.e. without the 24-instruction
//    renormalisation of -Y1 the bound L1 L2 + L3 L4 <= 6 used to demand;
//  * the full "PP == 0 (mod p)" comparison (27 compares over the limbs) runs only when limb 0 of PP is one of the three values it
//    can have then; P == acc (rare: equal points in one bucket) doubles the accumulator itself, so the operand is dead after the
//    first two products and they, too, run in place.
// a.x, a.y: the operand (y already negated where the digit is negative), consumed.
template <int VX, int VY> BB_HD void madd_ip(Xyzz& acc, bool& acc_inf, const Fe<Fq, 1, VX>& ax, const Fe<Fq, 1, VY>& ay)
{
#ifdef BBGPU_MADD_NO_ADDHI // round 3's form, kept for A/B builds
    auto U2 = mul_ip(ax, acc.zz);
    auto S2 = mul_ip(ay, acc.zzz);
    auto P = weak(sub(U2, acc.x));
    auto R = weak(sub(S2, acc.y));
#else
    // round 4: the two differences come out of their products' own reductions -- P = REDC(x2 ZZ1 + (K p - X1) 2^261), nine more additions inside the
    // carry chain instead of a subtraction (18 instructions) and a renormalisation (24) behind it -- with exact limbs
    auto P = mul_addhi_ip(ax, acc.zz, neg(acc.x));
    auto R = mul_addhi_ip(ay, acc.zzz, neg(acc.y));
#endif
    auto PP = sqr(P);
    constexpr Limbs9 p2 = make_multiple<Fq>(2);
    const bool same_x = (PP.d[0] == 0 || PP.d[0] == Fq::P[0] || PP.d[0] == p2.d[0]) && is_zero_mulout(PP); // rare
    // Two one-sided branches instead of if / else: each updates the accumulator's registers in place under its own lane mask.
    // An if / else is a two-way merge of the accumulator, which the compiler resolves with a second register set and 36 copies per
    // trip; the empty asm statement keeps it from folding the two back into one.
    if (same_x) {
        if (is_zero_slow(R)) {
            Xyzz d;
            dbl(d, acc);
            acc = d;
        } else {
            set_infinity(acc);
            acc_inf = true;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
    if (!same_x) {
        auto ZZ3 = mul_ip(acc.zz, PP);
        auto PPP = mul_ip(P, PP);
        auto Q = mul_ip(PP, acc.x);
#ifdef BBGPU_MADD_NO_ADDHI
        auto X3 = carry_full(sub(sqr(R), add(PPP, dbl(Q))));
#else
        auto X3 = sqr_addhi(R, neg(add(PPP, dbl(Q)))); // R^2 - (PPP + 2 Q) with exact limbs, no carry_full
#endif
        auto ZZZ3 = mul_ip(acc.zzz, PPP);
        auto Y3 = mul_add_ip(R, sub(Q, X3), neg(acc.y), PPP);
        acc.x = X3;
        acc.y = Y3;
        acc.zz = ZZ3;
        acc.zzz = ZZZ3;
    }
}

// r = p + q   [add-2008-s], 12M + 2S, all exceptional cases
BB_HD void add(Xyzz& r, const Xyzz& p, const Xyzz& q)
{
    const bool pinf = is_infinity(p), qinf = is_infinity(q);
    if (pinf) {
        r = q;
        return;
    }
    if (qinf) {
        r = p;
        return;
    }
    auto U1 = mul(p.x, q.zz);
    auto U2 = mul(q.x, p.zz);
    auto S1 = mul(p.y, q.zzz);
    auto S2 = mul(q.y, p.zzz);
    auto P = weak(sub(U2, U1));
    auto R = weak(sub(S2, S1));
    auto PP = sqr(P);
    if (is_zero_mulout(PP)) {
        if (is_zero_slow(R)) {
            dbl(r, p);
        } else {
            set_infinity(r);
        }
        return;
    }
    auto PPP = mul(P, PP);
    auto Q = mul(U1, PP);
    auto X3 = weak(sub(sqr(R), add(PPP, dbl(Q))));
    auto Y3 = mul_sub(R, sub(Q, X3), S1, PPP);
    auto ZZ3 = mul(mul(p.zz, q.zz), PP);
    auto ZZZ3 = mul(mul(p.zzz, q.zzz), PPP);
    r.x = X3;
    r.y = Y3;
    r.zz = ZZ3;
    r.zzz = ZZZ3;
}

// y -> -y when flag (branch-free select; the reference does this with cmov: group_impl_asm.tcc:71-153)
template <int V> BB_HD AffineV<V + 1> cond_neg_affine(const AffineV<V>& a, bool flag)
{
    AffineV<V + 1> r;
    r.x = a.x;
    Fe<Fq, 1, V + 1> ny = weak(neg(a.y));
#pragma unroll
    for (int i = 0; i < NL; i++) r.y.d[i] = flag ? ny.d[i] : a.y.d[i];
    return r;
}

// ---- memory formats ---------------------------------------------------------------------------------------------
// reference affine point: x,y as 4 x u64 Montgomery(2^256), 64 bytes (group.hpp:17-21)
BB_HD void load_affine_m256(AffineV<2>& r, const uint32_t* w16)
{
    uint32_t wx[8], wy[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        wx[i] = w16[i];
        wy[i] = w16[8 + i];
    }
    r.x = m256_to_m261<Fq>(unpack<Fq>(wx));
    r.y = m256_to_m261<Fq>(unpack<Fq>(wy));
}
// device-resident point (Montgomery-261, CANONICAL, packed 8 words per coordinate)
BB_HD void load_affine_m261(AffineV<1>& r, const uint32_t* w16)
{
    uint32_t wx[8], wy[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        wx[i] = w16[i];
        wy[i] = w16[8 + i];
    }
    Fe<Fq, 1, 6> ux = unpack<Fq>(wx), uy = unpack<Fq>(wy);
#pragma unroll
    for (int i = 0; i < NL; i++) {  // stored canonical by construction (store_affine_m261): value < p
        r.x.d[i] = ux.d[i];
        r.y.d[i] = uy.d[i];
    }
}
// The same with y -> p - y where `negy` (a negative digit), done on the PACKED words: one 8-word borrow chain and 8 selects, after
// which the ordinary unpack delivers exact limbs again.  (Negating the unpacked limbs costs as much -- 9 subtractions, 9 selects --
// and leaves y at L = 3, which the accumulator's y must not have: madd_ip relies on L = 1 there.)  y != 0 on this curve, so
// p - y is canonical; the type still says < 2p.
BB_HD void load_affine_m261_signed(Fe<Fq, 1, 1>& x, Fe<Fq, 1, 2>& y, const uint32_t (&w16)[16], bool negy)
{
    uint32_t wx[8], wy[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        wx[i] = w16[i];
        const uint32_t pi = (uint32_t)(Fq::P64[i >> 1] >> ((i & 1) * 32));
        const uint32_t yi = w16[8 + i];
        const uint32_t d = pi - yi, d2 = d - borrow;
        borrow = (uint32_t)(pi < yi) | (uint32_t)(d < borrow);
        wy[i] = negy ? d2 : yi;
    }
    Fe<Fq, 1, 6> ux = unpack<Fq>(wx), uy = unpack<Fq>(wy);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        x.d[i] = ux.d[i];
        y.d[i] = uy.d[i];
    }
}
BB_HD void store_affine_m261(uint32_t* w16, const Fe<Fq, 1, 12>& x, const Fe<Fq, 1, 12>& y)
{
    uint32_t w[8];
    to_canonical(x, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w16[i] = w[i];
    to_canonical(y, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w16[8 + i] = w[i];
}
// XYZZ <-> 4 x 8 words, Montgomery-261, values canonicalised so that they fit 256 bits
BB_HD void store_xyzz(uint32_t* w32, const Xyzz& p)
{
    uint32_t w[8];
    to_canonical(p.x, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[i] = w[i];
    to_canonical(p.y, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[8 + i] = w[i];
    to_canonical(p.zz, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[16 + i] = w[i];
    to_canonical(p.zzz, w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[24 + i] = w[i];
}
BB_HD void load_xyzz(Xyzz& p, const uint32_t* w32)
{
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = w32[i];
    p.x = unpack<Fq>(w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = w32[8 + i];
    p.y = unpack<Fq>(w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = w32[16 + i];
    p.zz = unpack<Fq>(w);
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = w32[24 + i];
    p.zzz = unpack<Fq>(w);
}
// XYZZ -> reference-format Jacobian-compatible triple in Montgomery(2^256): (X*ZZ? no) we emit the XYZZ
// coordinates converted to the reference's Montgomery form, 4 x 8 words; the host finishes (host_g1.hpp).
BB_HD void store_xyzz_m256(uint32_t* w32, const Xyzz& p)
{
    uint32_t w[8];
    to_canonical(m261_to_m256<Fq>(p.x), w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[i] = w[i];
    to_canonical(m261_to_m256<Fq>(p.y), w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[8 + i] = w[i];
    to_canonical(m261_to_m256<Fq>(p.zz), w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[16 + i] = w[i];
    to_canonical(m261_to_m256<Fq>(p.zzz), w);
#pragma unroll
    for (int i = 0; i < 8; i++) w32[24 + i] = w[i];
}

} // namespace bbgpu
