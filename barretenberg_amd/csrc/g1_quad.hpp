// g1_quad.hpp -- one XYZZ addition spread over the FOUR lanes of a quad (device only).
//
// The tail of an MSM (row / column sums, bit-sliced sums: csrc/msm.hip K5) is a chain of ~16 DEPENDENT group additions; a lone
// wave needs ~9 us for one (14 multiplications back to back on every lane, most lanes idle in the upper tree levels).  Here lane
// l of a quad holds coordinate l of a point (0: X, 1: Y, 2: ZZ, 3: ZZZ) and the addition [add-2008-s] runs as FOUR multiplication
// steps instead of fourteen, operands moving between the lanes with DPP quad permutes (register-to-register, no LDS):
//
//   step 1   M1 = P_l * Q_(l^2)                      -> U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1
//            D  = (U2 - U1 | S2 - S1 | U2 - U1 | S2 - S1) = (P | R | P | R)
//   step 2   N2 = (D D | D D | ZZ1 ZZ2 | ZZZ1 ZZZ2) -> PP | RR | ZZ12 | ZZZ12
//   step 3   N3 = (U1 | P | ZZ12 | P) * PP           -> Q | PPP | ZZ3 | PPP
//            X3 = RR - PPP - 2Q                         (lane 1)
//   step 4   N4 = (S1 PPP | R (Q - X3) | - | ZZZ12 PPP)  ;  Y3 = N4[1] - N4[0]   ;  ZZZ3 = N4[3]
//
// Every lane executes the same instruction stream (one multiplication per step; the operands are chosen by lane), so the cost per
// addition is ~4 multiplications + ~100 DPP moves + selects instead of 14 multiplications: ~1,250 instructions against ~3,700.
// Exceptional cases as in g1.hpp: an infinite operand (zz == 0, decided by lane 2, broadcast) returns the other; equal x (PP == 0,
// decided by lane 0) is rare and handled by gathering both points into every lane and running the ordinary add().
// Semantics: g1::add (src/barretenberg/groups/group.hpp:324-448), any representative being legal before normalisation.
#pragma once
#include "g1.hpp"

namespace bbgpu {

constexpr int QP_XOR2 = 0x4E; // quad_perm [2, 3, 0, 1]
constexpr int QP_B0 = 0x00;   // [0, 0, 0, 0]
constexpr int QP_B1 = 0x55;   // [1, 1, 1, 1]
constexpr int QP_B2 = 0xAA;   // [2, 2, 2, 2]
constexpr int QP_B3 = 0xFF;   // [3, 3, 3, 3]
constexpr int QP_0022 = 0xA0; // [0, 0, 2, 2]

template <int CTRL> __device__ __forceinline__ uint32_t qperm(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL, class F, int L, int V> __device__ __forceinline__ Fe<F, L, V> qperm(const Fe<F, L, V>& a)
{
    Fe<F, L, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = qperm<CTRL>(a.d[i]);
    return r;
}
template <class F, int L, int V> __device__ __forceinline__ Fe<F, L, V> qsel(bool c, const Fe<F, L, V>& a, const Fe<F, L, V>& b)
{
    Fe<F, L, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.d[i] = c ? a.d[i] : b.d[i];
    return r;
}
template <int L2, int V2, class F, int L, int V> __device__ __forceinline__ Fe<F, L2, V2> widen(const Fe<F, L, V>& a)
{
    return Fe<F, L2, V2>(a); // the converting constructor checks L <= L2, V <= V2
}

// the rare equal-x case: gather both points into every lane and run the ordinary addition.  NOT inlined: its ~150 live registers
// must not weigh on the register allocation of the four-step path around it.
__device__ __noinline__ FqN quad_add_by_gather(const FqN p, const FqN q, const uint32_t l)
{
    Xyzz pf, qf, rf;
    pf.x = qperm<QP_B0>(p); pf.y = qperm<QP_B1>(p); pf.zz = qperm<QP_B2>(p); pf.zzz = qperm<QP_B3>(p);
    qf.x = qperm<QP_B0>(q); qf.y = qperm<QP_B1>(q); qf.zz = qperm<QP_B2>(q); qf.zzz = qperm<QP_B3>(q);
    add(rf, pf, qf);
    return l == 0 ? rf.x : (l == 1 ? rf.y : (l == 2 ? rf.zz : rf.zzz));
}

// this lane's coordinate (lane & 3) of p + q, given this lane's coordinate of p and of q
__device__ __forceinline__ FqN quad_add(const FqN& p, const FqN& q, uint32_t l)
{
    const bool pinf = qperm<QP_B2>(limbs_all_zero(p) ? 1u : 0u) != 0; // zz lives on lane 2
    const bool qinf = qperm<QP_B2>(limbs_all_zero(q) ? 1u : 0u) != 0;
    const bool lo = l < 2;
    // step 1
    const auto m1 = mul(p, qperm<QP_XOR2>(q));              // U1 | S1 | U2 | S2
    const auto d1 = qperm<QP_XOR2>(m1);
    const auto D = weak(sub(qsel(lo, d1, m1), qsel(lo, m1, d1))); // P | R | P | R        < 5p
    // step 2
    const FqN Dw = widen<1, 12>(D);
    const auto n2 = mul(qsel(lo, Dw, p), qsel(lo, Dw, q));  // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const bool same_x = qperm<QP_B0>(is_zero_mulout(n2) ? 1u : 0u) != 0 && !pinf && !qinf;
    // step 3
    const auto ppb = qperm<QP_B0>(n2);                      // PP on every lane
    const auto pD = qperm<QP_0022>(D);                      // lanes 1, 3 receive P
    const Fe<Fq, 1, 5> a3 = l == 0 ? widen<1, 5>(m1) : (l == 2 ? widen<1, 5>(n2) : pD);
    const auto n3 = mul(a3, ppb);                           // Q | PPP | ZZ3 | PPP
    // step 4
    const auto qb = qperm<QP_B0>(n3);                       // Q on every lane
    const auto x3 = weak(sub(n2, add(n3, dbl(qb))));        // lane 1: RR - PPP - 2Q             < 9p
    const FqN qmx = weak(sub(qb, x3));                      // Q - X3                            < 12p
    const auto s1b = qperm<QP_B1>(m1), pppb = qperm<QP_B1>(n3);
    const FqN a4 = l == 1 ? Dw : (l == 3 ? widen<1, 12>(n2) : widen<1, 12>(s1b));
    const FqN b4 = l == 1 ? qmx : (l == 3 ? widen<1, 12>(n3) : widen<1, 12>(pppb));
    const auto n4 = mul(a4, b4);                            // S1 PPP | R (Q - X3) | - | ZZZ3
    const auto y3 = weak(sub(n4, qperm<QP_B0>(n4)));        // lane 1: Y3                        < 5p
    const auto x3b = qperm<QP_B1>(x3);                      // X3 to lane 0
    FqN r = l == 0 ? widen<1, 12>(x3b) : (l == 1 ? widen<1, 12>(y3) : (l == 2 ? widen<1, 12>(n3) : widen<1, 12>(n4)));
    if (__any(same_x ? 1 : 0)) { // rare: P == +-Q in some quad of this wave
        const FqN g = quad_add_by_gather(p, q, l);
        if (same_x) r = g;
    }
    if (pinf) r = q;
    else if (qinf) r = p;
    return r;
}

} // namespace bbgpu
