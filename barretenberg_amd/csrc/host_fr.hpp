// host_fr.hpp -- host-side BN254 scalar-field (fr) arithmetic for the O(1)-sized scalar work of the resident PLONK prover
// (Fiat-Shamir challenges, powers of challenges, a handful of inversions) and for preparing kernel constants.
// 4 x 64-bit Montgomery, R = 2^256: the reference's own memory format (fields/field.hpp:19-22), so values can be
// hashed / written into the proof as they are.  Semantics restated from field_impl_int128.tcc:72-137,149-263
// (Montgomery product followed by one conditional subtraction; canonical in, canonical out).
// This is product code: it does NOT use oracle/.
#pragma once
#include <stdint.h>
#include <string.h>

#include "bn254_params.h"
#include "fe.hpp"
#include "host_modinv.hpp"

namespace bbgpu {
namespace host {

typedef unsigned __int128 u128;

struct Fr {
    uint64_t d[4];
};

static inline Fr fr_from_limbs(const uint64_t (&w)[4])
{
    Fr r;
    memcpy(r.d, w, 32);
    return r;
}
static inline Fr fr_zero() { Fr r; memset(r.d, 0, 32); return r; }
static inline Fr fr_one() { return fr_from_limbs(FrHostP::ONE); }
static inline bool fr_is_zero(const Fr& a) { return (a.d[0] | a.d[1] | a.d[2] | a.d[3]) == 0; }
static inline bool fr_eq(const Fr& a, const Fr& b) { return !memcmp(a.d, b.d, 32); }

static inline void fr_cond_sub_p(Fr& a)
{
    uint64_t t[4];
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a.d[i] - FrHostP::P[i] - (uint64_t)br;
        t[i] = (uint64_t)s;
        br = (s >> 64) & 1;
    }
    if (!br) memcpy(a.d, t, 32);
}
static inline Fr fr_add(const Fr& a, const Fr& b)
{
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.d[i] + b.d[i];
        r.d[i] = (uint64_t)c;
        c >>= 64;
    }
    fr_cond_sub_p(r); // a + b < 2r < 2^256
    return r;
}
static inline Fr fr_sub(const Fr& a, const Fr& b)
{
    Fr r;
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a.d[i] - b.d[i] - (uint64_t)br;
        r.d[i] = (uint64_t)s;
        br = (s >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.d[i] + FrHostP::P[i];
            r.d[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
static inline Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }
// a may be any 256-bit value as long as b < r (used to reduce a raw 256-bit hash: field.hpp:224-232 semantics)
static inline Fr fr_mul(const Fr& a, const Fr& b)
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
        uint64_t m = t[0] * FrHostP::PINV;
        c = (u128)m * FrHostP::P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FrHostP::P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t4;
        t[3] = (uint64_t)c;
        t[4] = t5 + (uint64_t)(c >> 64);
    }
    Fr r = { { t[0], t[1], t[2], t[3] } };
    fr_cond_sub_p(r);
    return r;
}
static inline Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }
static inline Fr fr_pow(const Fr& a, uint64_t e)
{
    Fr acc = fr_one(), b = a;
    for (; e; e >>= 1) {
        if (e & 1) acc = fr_mul(acc, b);
        b = fr_sqr(b);
    }
    return acc;
}
static inline Fr fr_inv_fermat(const Fr& a) // a^(r-2); inv(0) = 0 -- the reference's route (field.hpp:258-348), kept as the check of fr_inv
{
    const uint64_t e[4] = { FrHostP::P[0] - 2, FrHostP::P[1], FrHostP::P[2], FrHostP::P[3] };
    Fr acc = fr_one();
    for (int i = 255; i >= 0; --i) {
        acc = fr_sqr(acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a);
    }
    return acc;
}
// Montgomery in, Montgomery out, inv(0) = 0: (x R)^-1 = x^-1 R^-1 by divsteps (host_modinv.hpp: ~1.5 us against ~10 us), times R^3 = RSQ * RSQ / R
static inline Fr fr_inv(const Fr& a)
{
    static const ModInfo info = modinfo_from(FrHostP::P);
    static const Fr r3 = fr_mul(fr_from_limbs(FrHostP::RSQ), fr_from_limbs(FrHostP::RSQ));
    Fr t;
    modinv_u64x4(a.d, info, t.d);
    return fr_mul(t, r3);
}
// plain integer (any 256-bit value) -> Montgomery form, reduced mod r
static inline Fr fr_to_mont(const Fr& raw) { return fr_mul(raw, fr_from_limbs(FrHostP::RSQ)); }
static inline Fr fr_from_mont(const Fr& a)
{
    Fr one_raw = fr_zero();
    one_raw.d[0] = 1;
    return fr_mul(a, one_raw);
}
static inline Fr fr_from_u64(uint64_t v)
{
    Fr raw = fr_zero();
    raw.d[0] = v;
    return fr_to_mont(raw);
}
// primitive 2^k-th root of unity the reference's evaluation_domain uses (field.hpp:487-494 from fr.hpp:59-63)
static inline Fr fr_root_of_unity(int log2n)
{
    Fr r = fr_from_limbs(FrHostP::ROOT28);
    for (int i = 28; i > log2n; --i) r = fr_sqr(r);
    return r;
}

// ---- kernel constants: canonical 9 x 29-bit limbs ----------------------------------------------------------------------
// the same residue the caller holds (x * 2^256), just re-limbed: adds to / subtracts from memory-format values on the device
static inline Limbs9 limbs_m256(const Fr& a)
{
    uint32_t w[8];
    for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)a.d[i]; w[2 * i + 1] = (uint32_t)(a.d[i] >> 32); }
    Fe<FrP, 1, 6> u = unpack<FrP>(w);
    Limbs9 r;
    for (int i = 0; i < NL; i++) r.d[i] = u.d[i];
    return r;
}
// x * 2^261: a multiplier for memory-format values (mont261(a * 2^256, x * 2^261) = a x * 2^256)
static inline Limbs9 limbs_m261(const Fr& a) { return limbs_m256(fr_mul(a, fr_from_limbs(FrHostP::M32))); }

} // namespace host
} // namespace bbgpu
