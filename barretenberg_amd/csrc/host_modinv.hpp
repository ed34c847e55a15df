// host_modinv.hpp -- modular inversion by Bernstein-Yang divsteps ("safegcd", eprint 2019/266) for the two BN254 moduli, host only.
// One Fermat chain (a^(p-2): 254 squarings + ~128 multiplications = ~10 us with the 25-ns host multiplication) stands between the last
// kernel of every MSM and its normalised result; this takes ~1.5 us.  Structure: batches of 62 divsteps on the low 64 bits of (f, g)
// produce a 2x2 transition matrix scaled by 2^62, which is applied to the full-width (f, g) (exact division by 2^62) and to (d, e) modulo p
// (division by 2^62 mod p: add the multiple of p that clears the low 62 bits); 12 batches cover the 741 divsteps that suffice for 256-bit
// inputs from delta = 1, and the loop stops as soon as g = 0.  Values are 5 signed limbs of 62 bits.  Variable time (public data only: the
// denominators of commitments).  Checked against the Fermat inversion on random and edge values (tests/cpp/test_host_sanitize.cpp).
#pragma once
#include <stdint.h>
#include <string.h>

namespace bbgpu {
namespace host {

struct S62 {
    int64_t v[5];
};
static const int64_t MODINV_M62 = (int64_t)(UINT64_MAX >> 2);

static inline S62 s62_from_u64x4(const uint64_t a[4])
{
    S62 r;
    r.v[0] = (int64_t)(a[0] & (uint64_t)MODINV_M62);
    r.v[1] = (int64_t)(((a[0] >> 62) | (a[1] << 2)) & (uint64_t)MODINV_M62);
    r.v[2] = (int64_t)(((a[1] >> 60) | (a[2] << 4)) & (uint64_t)MODINV_M62);
    r.v[3] = (int64_t)(((a[2] >> 58) | (a[3] << 6)) & (uint64_t)MODINV_M62);
    r.v[4] = (int64_t)(a[3] >> 56);
    return r;
}
// value must be in [0, 2^256)
static inline void s62_to_u64x4(const S62& a, uint64_t out[4])
{
    const uint64_t v0 = (uint64_t)a.v[0], v1 = (uint64_t)a.v[1], v2 = (uint64_t)a.v[2], v3 = (uint64_t)a.v[3], v4 = (uint64_t)a.v[4];
    out[0] = v0 | (v1 << 62);
    out[1] = (v1 >> 2) | (v2 << 60);
    out[2] = (v2 >> 4) | (v3 << 58);
    out[3] = (v3 >> 6) | (v4 << 56);
}

struct ModInfo {
    S62 modulus;
    uint64_t modulus_inv62; // modulus^-1 mod 2^62
};
static inline ModInfo modinfo_from(const uint64_t p[4])
{
    ModInfo m;
    m.modulus = s62_from_u64x4(p);
    uint64_t x = p[0]; // Newton: x <- x (2 - p x), doubling the number of correct low bits (p odd: p * p = 1 mod 8)
    for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
    m.modulus_inv62 = x & (uint64_t)MODINV_M62;
    return m;
}

struct Trans {
    int64_t u, v, q, r;
};
// 62 divsteps on the low bits; on return 2^62 (f', g') = [[u, v], [q, r]] (f, g)
static inline int64_t divsteps_62(int64_t delta, uint64_t f0, uint64_t g0, Trans* t)
{
    uint64_t u = 1, v = 0, q = 0, r = 1; // two's complement arithmetic on purpose
    uint64_t f = f0, g = g0;
    for (int i = 0; i < 62; i++) {
        if (g & 1) {
            if (delta > 0) {
                const uint64_t nf = g, ng = g - f, nu = q, nv = r, nq = q - u, nr = r - v;
                f = nf; g = ng; u = nu; v = nv; q = nq; r = nr;
                delta = -delta;
            } else {
                g += f; q += u; r += v;
            }
        }
        g = (uint64_t)((int64_t)g >> 1);
        u <<= 1;
        v <<= 1;
        delta += 1;
    }
    t->u = (int64_t)u; t->v = (int64_t)v; t->q = (int64_t)q; t->r = (int64_t)r;
    return delta;
}
typedef __int128 i128;
// (f, g) <- [[u, v], [q, r]] (f, g) / 2^62, exact
static inline void update_fg(S62& f, S62& g, const Trans& t)
{
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    i128 cf = (i128)u * f.v[0] + (i128)v * g.v[0];
    i128 cg = (i128)q * f.v[0] + (i128)r * g.v[0];
    cf >>= 62; // the low 62 bits are zero by construction
    cg >>= 62;
    for (int i = 1; i < 5; i++) {
        cf += (i128)u * f.v[i] + (i128)v * g.v[i];
        cg += (i128)q * f.v[i] + (i128)r * g.v[i];
        const int64_t nf = (int64_t)((uint64_t)cf & (uint64_t)MODINV_M62), ng = (int64_t)((uint64_t)cg & (uint64_t)MODINV_M62);
        f.v[i - 1] = nf;
        g.v[i - 1] = ng;
        cf >>= 62;
        cg >>= 62;
    }
    f.v[4] = (int64_t)cf;
    g.v[4] = (int64_t)cg;
}
// (d, e) <- [[u, v], [q, r]] (d, e) / 2^62 modulo p; d, e stay in (-2p, p)
static inline void update_de(S62& d, S62& e, const Trans& t, const ModInfo& m)
{
    const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int64_t sd = d.v[4] >> 63, se = e.v[4] >> 63; // all ones for a negative value
    int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se); // the multiples of p that bring negative inputs back first
    i128 cd = (i128)u * d.v[0] + (i128)v * e.v[0];
    i128 ce = (i128)q * d.v[0] + (i128)r * e.v[0];
    md -= (int64_t)((m.modulus_inv62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)MODINV_M62);
    me -= (int64_t)((m.modulus_inv62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)MODINV_M62);
    cd += (i128)m.modulus.v[0] * md;
    ce += (i128)m.modulus.v[0] * me;
    cd >>= 62; // low 62 bits cleared by the choice of md, me
    ce >>= 62;
    for (int i = 1; i < 5; i++) {
        cd += (i128)u * d.v[i] + (i128)v * e.v[i] + (i128)m.modulus.v[i] * md;
        ce += (i128)q * d.v[i] + (i128)r * e.v[i] + (i128)m.modulus.v[i] * me;
        d.v[i - 1] = (int64_t)((uint64_t)cd & (uint64_t)MODINV_M62);
        e.v[i - 1] = (int64_t)((uint64_t)ce & (uint64_t)MODINV_M62);
        cd >>= 62;
        ce >>= 62;
    }
    d.v[4] = (int64_t)cd;
    e.v[4] = (int64_t)ce;
}
static inline bool s62_is_zero(const S62& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4]) == 0; }
// r <- (negate ? -a : a) brought into [0, p); a in (-2p, p)
static inline void s62_normalise(S62& a, bool negate, const ModInfo& m)
{
    auto add_p = [&](int64_t mask) { // a += p & mask (mask = 0 or -1), then carry
        for (int i = 0; i < 5; i++) a.v[i] += m.modulus.v[i] & mask;
    };
    auto carry = [&]() {
        for (int i = 0; i < 4; i++) {
            a.v[i + 1] += a.v[i] >> 62;
            a.v[i] &= MODINV_M62;
        }
    };
    carry();
    add_p(a.v[4] >> 63); // negative -> + p  (now in (-p, p))
    carry();
    if (negate)
        for (int i = 0; i < 5; i++) a.v[i] = -a.v[i];
    carry();
    add_p(a.v[4] >> 63);
    carry();
    add_p(a.v[4] >> 63);
    carry();
}
// x^-1 mod p as integers (x in [0, p), p odd prime); 0 -> 0
static inline void modinv_u64x4(const uint64_t x[4], const ModInfo& m, uint64_t out[4])
{
    S62 d = { { 0, 0, 0, 0, 0 } }, e = { { 1, 0, 0, 0, 0 } };
    S62 f = m.modulus, g = s62_from_u64x4(x);
    int64_t delta = 1;
    for (int batch = 0; batch < 12 && !s62_is_zero(g); batch++) {
        Trans t;
        delta = divsteps_62(delta, (uint64_t)f.v[0], (uint64_t)g.v[0], &t);
        update_de(d, e, t, m);
        update_fg(f, g, t);
    }
    // g = 0, f = +-gcd = +-1 (or +-p for x = 0: d is then 0 mod p)
    s62_normalise(d, f.v[4] < 0, m);
    s62_to_u64x4(d, out);
}

} // namespace host
} // namespace bbgpu
