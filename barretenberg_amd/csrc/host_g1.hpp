// host_g1.hpp -- host-side BN254 fq / G1 arithmetic for the O(1)-sized tail of an MSM (the final Horner fold over
// windows, summing per-rank partial sums, and the single normalisation).  4 x 64-bit Montgomery (R = 2^256), the
// reference's own memory format (fields/field.hpp:19-22), so results can be handed straight back to the caller.
// This is product code: it does NOT use oracle/.  Semantics restated from field_impl_int128.tcc:72-137,248-255
// (Montgomery product with one final conditional subtraction); point formulas are the standard XYZZ ones used on the
// device (g1.hpp), any representative being legal before normalisation.
#pragma once
#include <stdint.h>
#include <string.h>

#include "host_modinv.hpp"

namespace bbgpu {
namespace host {

typedef unsigned __int128 u128;

struct Fq {
    uint64_t d[4];
};

static const uint64_t FQ_P[4] = { 0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL };
static const uint64_t FQ_PINV = 0x87d20782e4866389ULL; // -p^-1 mod 2^64 (fq.hpp:64)
static const Fq FQ_ONE = { { 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL } }; // fq.hpp:33-36

static inline bool fq_is_zero(const Fq& a) { return (a.d[0] | a.d[1] | a.d[2] | a.d[3]) == 0; }
static inline bool fq_eq(const Fq& a, const Fq& b) { return !memcmp(a.d, b.d, 32); }

static inline void fq_cond_sub_p(Fq& a)
{
    uint64_t t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a.d[i] - FQ_P[i] - (uint64_t)br;
        t[i] = (uint64_t)s;
        br = (s >> 64) & 1;
    }
    if (!br) memcpy(a.d, t, 32);
}
// canonical in, canonical out
static inline Fq fq_add(const Fq& a, const Fq& b)
{
    Fq r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.d[i] + b.d[i];
        r.d[i] = (uint64_t)c;
        c >>= 64;
    }
    fq_cond_sub_p(r); // a + b < 2p < 2^256
    return r;
}
static inline Fq fq_sub(const Fq& a, const Fq& b)
{
    Fq r;
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a.d[i] - b.d[i] - (uint64_t)br;
        r.d[i] = (uint64_t)s;
        br = (s >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.d[i] + FQ_P[i];
            r.d[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
static inline Fq fq_mul(const Fq& a, const Fq& b)
{
    uint64_t t[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.d[i] * b.d[j] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        uint64_t t4 = (uint64_t)c, t5 = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FQ_PINV;
        c = (u128)m * FQ_P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FQ_P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t4;
        t[3] = (uint64_t)c;
        t[4] = t5 + (uint64_t)(c >> 64);
    }
    Fq r = { { t[0], t[1], t[2], t[3] } };
    fq_cond_sub_p(r);
    return r;
}
static inline Fq fq_sqr(const Fq& a) { return fq_mul(a, a); }
static inline Fq fq_dbl(const Fq& a) { return fq_add(a, a); }
static inline Fq fq_inv_fermat(const Fq& a) // a^(p-2): the reference's own route (field.hpp:258-348), kept as the check of fq_inv
{
    uint64_t e[4] = { FQ_P[0] - 2, FQ_P[1], FQ_P[2], FQ_P[3] };
    Fq acc = FQ_ONE;
    for (int i = 255; i >= 0; --i) {
        acc = fq_sqr(acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = fq_mul(acc, a);
    }
    return acc;
}

// Montgomery in, Montgomery out: (x R)^-1 = x^-1 R^-1 by divsteps (host_modinv.hpp, ~1.5 us against ~10 us for the Fermat chain), times R^3
static inline Fq fq_inv(const Fq& a)
{
    static const ModInfo info = modinfo_from(FQ_P);
    static const Fq r3 = [] {
        Fq r2 = FQ_ONE; // R mod p, doubled 256 times = R^2 mod p
        for (int i = 0; i < 256; i++) r2 = fq_dbl(r2);
        return fq_mul(r2, r2); // R^4 / R = R^3
    }();
    Fq t;
    modinv_u64x4(a.d, info, t.d);
    return fq_mul(t, r3);
}

struct Xyzz {
    Fq x, y, zz, zzz; // infinity <=> zz == 0
};
static inline Xyzz g1_infinity()
{
    Xyzz r;
    memset(&r, 0, sizeof(r));
    return r;
}
static inline bool g1_is_inf(const Xyzz& p) { return fq_is_zero(p.zz); }

static inline Xyzz g1_dbl(const Xyzz& p)
{
    if (g1_is_inf(p)) return p;
    Fq U = fq_dbl(p.y), V = fq_sqr(U), W = fq_mul(U, V), S = fq_mul(p.x, V), XX = fq_sqr(p.x);
    Fq M = fq_add(fq_dbl(XX), XX);
    Xyzz r;
    r.x = fq_sub(fq_sqr(M), fq_dbl(S));
    r.y = fq_sub(fq_mul(M, fq_sub(S, r.x)), fq_mul(W, p.y));
    r.zz = fq_mul(V, p.zz);
    r.zzz = fq_mul(W, p.zzz);
    return r;
}
static inline Xyzz g1_add(const Xyzz& p, const Xyzz& q)
{
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    Fq U1 = fq_mul(p.x, q.zz), U2 = fq_mul(q.x, p.zz), S1 = fq_mul(p.y, q.zzz), S2 = fq_mul(q.y, p.zzz);
    Fq P = fq_sub(U2, U1), R = fq_sub(S2, S1);
    if (fq_is_zero(P)) return fq_is_zero(R) ? g1_dbl(p) : g1_infinity();
    Fq PP = fq_sqr(P), PPP = fq_mul(P, PP), Q = fq_mul(U1, PP);
    Xyzz r;
    r.x = fq_sub(fq_sub(fq_sqr(R), PPP), fq_dbl(Q));
    r.y = fq_sub(fq_mul(R, fq_sub(Q, r.x)), fq_mul(S1, PPP));
    r.zz = fq_mul(fq_mul(p.zz, q.zz), PP);
    r.zzz = fq_mul(fq_mul(p.zzz, q.zzz), PPP);
    return r;
}
// reference Jacobian {x,y,z} (group.hpp:23-28) <-> XYZZ: zz = z^2, zzz = z^3
static inline Xyzz g1_from_jacobian(const uint64_t j[12])
{
    if ((j[7] >> 63) & 1) return g1_infinity();
    Xyzz r;
    Fq z;
    memcpy(r.x.d, j, 32);
    memcpy(r.y.d, j + 4, 32);
    memcpy(z.d, j + 8, 32);
    // callers hand canonical coordinates (normalised outputs, or reference results which are < p)
    r.zz = fq_sqr(z);
    r.zzz = fq_mul(r.zz, z);
    return r;
}
// normalised reference element: x, y canonical, z = one; infinity: y msb set (group.hpp:133-151), rest zero
static inline void g1_to_normalised(const Xyzz& p, uint64_t out[12])
{
    memset(out, 0, 96);
    if (g1_is_inf(p)) {
        out[7] = 1ULL << 63;
        return;
    }
    Fq i = fq_inv(fq_mul(p.zz, p.zzz));
    Fq izz = fq_mul(i, p.zzz), izzz = fq_mul(i, p.zz);
    Fq x = fq_mul(p.x, izz), y = fq_mul(p.y, izzz);
    memcpy(out, x.d, 32);
    memcpy(out + 4, y.d, 32);
    memcpy(out + 8, FQ_ONE.d, 32);
}

// the same for several points with ONE inversion (Montgomery's trick over the denominators zz * zzz; fields/field.hpp:503-522 is the
// reference's batch_invert): a prover round's commitments come back together, and a Fermat inversion is ~11 us of host time each
static inline void g1_batch_to_normalised(const Xyzz* p, size_t count, uint64_t* out /* count x 12 */)
{
    Fq den[8], pre[8];
    if (count > 8) { // not a batch size the library produces: one by one
        for (size_t i = 0; i < count; i++) g1_to_normalised(p[i], out + 12 * i);
        return;
    }
    Fq run = FQ_ONE;
    for (size_t i = 0; i < count; i++) {
        pre[i] = run;
        den[i] = g1_is_inf(p[i]) ? FQ_ONE : fq_mul(p[i].zz, p[i].zzz);
        run = fq_mul(run, den[i]);
    }
    Fq inv = fq_inv(run);
    for (size_t i = count; i-- > 0;) {
        const Fq di = fq_mul(inv, pre[i]); // 1 / den[i]
        inv = fq_mul(inv, den[i]);
        uint64_t* o = out + 12 * i;
        memset(o, 0, 96);
        if (g1_is_inf(p[i])) {
            o[7] = 1ULL << 63;
            continue;
        }
        const Fq izz = fq_mul(di, p[i].zzz), izzz = fq_mul(di, p[i].zz);
        const Fq x = fq_mul(p[i].x, izz), y = fq_mul(p[i].y, izzz);
        memcpy(o, x.d, 32);
        memcpy(o + 4, y.d, 32);
        memcpy(o + 8, FQ_ONE.d, 32);
    }
}

} // namespace host
} // namespace bbgpu
