// host_g2.hpp -- just enough BN254 G2 arithmetic on the host to write the G2 half of an ignition-format transcript
// (io/io.hpp:100-135,171-180: two G2 points behind the G1 points, of which index 1 must be x * G2 -- the verifier's pairing input).
// fq2 = fq[u] / (u^2 + 1) (curves/bn254/fq2.hpp); G2 is the sextic twist y^2 = x^3 + 3 / (9 + u); generator curves/bn254/g2.hpp:14-21.
// One scalar multiplication per transcript: plain double-and-add in Jacobian coordinates (the curve constant b never enters the
// a = 0 formulas).  Product code, no oracle/; fq arithmetic from host_g1.hpp (4 x 64-bit Montgomery, the reference's memory format).
#pragma once
#include "host_fr.hpp"
#include "host_g1.hpp"

namespace bbgpu {
namespace host {

struct Fq2 {
    Fq c0, c1;
};
static inline Fq2 fq2_add(const Fq2& a, const Fq2& b) { return { fq_add(a.c0, b.c0), fq_add(a.c1, b.c1) }; }
static inline Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return { fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1) }; }
static inline Fq2 fq2_dbl(const Fq2& a) { return fq2_add(a, a); }
static inline Fq2 fq2_mul(const Fq2& a, const Fq2& b)
{
    const Fq t0 = fq_mul(a.c0, b.c0), t1 = fq_mul(a.c1, b.c1);
    const Fq cross = fq_sub(fq_sub(fq_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1)), t0), t1);
    return { fq_sub(t0, t1), cross }; // u^2 = -1
}
static inline Fq2 fq2_sqr(const Fq2& a) { return fq2_mul(a, a); }
static inline Fq2 fq2_inv(const Fq2& a)
{
    const Fq zero = { { 0, 0, 0, 0 } };
    const Fq norm_inv = fq_inv(fq_add(fq_sqr(a.c0), fq_sqr(a.c1))); // 1 / (c0^2 + c1^2)
    return { fq_mul(a.c0, norm_inv), fq_mul(fq_sub(zero, a.c1), norm_inv) };
}
static inline bool fq2_is_zero(const Fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }

struct G2Affine {
    Fq2 x, y;
};
struct G2Jac {
    Fq2 x, y, z; // infinity <=> z == 0
};
// g2::affine_one (g2.hpp:14-21, Montgomery form)
static const G2Affine G2_ONE = {
    { { { 0x8e83b5d102bc2026ULL, 0xdceb1935497b0172ULL, 0xfbb8264797811adfULL, 0x19573841af96503bULL } },
      { { 0xafb4737da84c6140ULL, 0x6043dd5a5802d8c4ULL, 0x09e950fc52a02f86ULL, 0x14fef0833aea7b6bULL } } },
    { { { 0x619dfa9d886be9f6ULL, 0xfe7fd297f59e9b78ULL, 0xff9e1a62231b7dfeULL, 0x28fd7eebae9e4206ULL } },
      { { 0x64095b56c71856eeULL, 0xdc57f922327d3cbbULL, 0x55f935be33351076ULL, 0x0da4a0e693fd6482ULL } } }
};

static inline G2Jac g2_dbl(const G2Jac& p) // dbl-2009-l, a = 0
{
    if (fq2_is_zero(p.z)) return p;
    const Fq2 A = fq2_sqr(p.x), B = fq2_sqr(p.y), C = fq2_sqr(B);
    const Fq2 D = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr(fq2_add(p.x, B)), A), C));
    const Fq2 E = fq2_add(fq2_dbl(A), A), F = fq2_sqr(E);
    G2Jac r;
    r.x = fq2_sub(F, fq2_dbl(D));
    r.y = fq2_sub(fq2_mul(E, fq2_sub(D, r.x)), fq2_dbl(fq2_dbl(fq2_dbl(C))));
    r.z = fq2_dbl(fq2_mul(p.y, p.z));
    return r;
}
static inline G2Jac g2_madd(const G2Jac& p, const G2Affine& q) // madd-2007-bl with the exceptional cases
{
    const Fq2 one = { FQ_ONE, { { 0, 0, 0, 0 } } };
    if (fq2_is_zero(p.z)) return { q.x, q.y, one };
    const Fq2 Z1Z1 = fq2_sqr(p.z), U2 = fq2_mul(q.x, Z1Z1), S2 = fq2_mul(fq2_mul(q.y, p.z), Z1Z1);
    const Fq2 H = fq2_sub(U2, p.x), rr = fq2_dbl(fq2_sub(S2, p.y));
    if (fq2_is_zero(H)) {
        if (fq2_is_zero(rr)) return g2_dbl({ q.x, q.y, one });
        return { one, one, { { { 0, 0, 0, 0 } }, { { 0, 0, 0, 0 } } } };
    }
    const Fq2 HH = fq2_sqr(H), I = fq2_dbl(fq2_dbl(HH)), J = fq2_mul(H, I), V = fq2_mul(p.x, I);
    G2Jac r;
    r.x = fq2_sub(fq2_sub(fq2_sqr(rr), J), fq2_dbl(V));
    r.y = fq2_sub(fq2_mul(rr, fq2_sub(V, r.x)), fq2_dbl(fq2_mul(p.y, J)));
    r.z = fq2_sub(fq2_sub(fq2_sqr(fq2_add(p.z, H)), Z1Z1), HH);
    return r;
}
// k * q for a scalar given in Montgomery form (any representative); k = 0 mod r is not a transcript secret: returns q's infinity as z = 0 -> caller rejects
static inline bool g2_scalar_mul_affine(const G2Affine& q, const Fr& k_mont, G2Affine* out)
{
    const Fr k = fr_from_mont(k_mont);
    G2Jac acc = { q.x, q.y, { { { 0, 0, 0, 0 } }, { { 0, 0, 0, 0 } } } };
    for (int i = 255; i >= 0; --i) {
        acc = g2_dbl(acc);
        if ((k.d[i >> 6] >> (i & 63)) & 1) acc = g2_madd(acc, q);
    }
    if (fq2_is_zero(acc.z)) return false;
    const Fq2 zi = fq2_inv(acc.z), zi2 = fq2_sqr(zi);
    out->x = fq2_mul(acc.x, zi2);
    out->y = fq2_mul(acc.y, fq2_mul(zi2, zi));
    return true;
}

} // namespace host
} // namespace bbgpu
