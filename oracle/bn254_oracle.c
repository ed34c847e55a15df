/*
 * bn254_oracle.c -- TEST INFRASTRUCTURE ONLY (see bn254_oracle.h).
 *
 * CPU restatement, in plain C with unsigned __int128, of the BN254 field, group,
 * Pippenger and radix-2 NTT algorithms of arielgabizon/barretenberg.  The step
 * order of every routine follows the reference's portable path so that results
 * agree limb-for-limb even for the out-of-range operands used by the
 * reference's own known-answer tests.  File:line citations are relative to
 * /root/reference/src/barretenberg/.
 *
 * Parity status: PINNED -- tests/test_oracle.py checks this file against the
 * KATs in test/test_fq.cpp, test/test_fr.cpp, test/test_g1.cpp and against
 * fixtures produced by the reference itself (oracle/_ref, tests/golden/).
 */
#include "bn254_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* parameters: curves/bn254/fq.hpp:12-64, curves/bn254/fr.hpp:12-81          */
/* ------------------------------------------------------------------------- */
struct fparams {
    uint64_t mod[4];
    uint64_t twice_mod[4];
    uint64_t one[4];       /* R mod p */
    uint64_t r_squared[4]; /* R^2 mod p */
    uint64_t beta[4];      /* cube root of unity (Montgomery) */
    uint64_t r_inv;        /* -p^-1 mod 2^64 */
};

static const struct fparams FQ = {
    { 0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL },
    { 0x7841182db0f9fa8eULL, 0x2f02d522d0e3951aULL, 0x70a08b6d0302b0bbULL, 0x60c89ce5c2634053ULL },
    { 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL },
    { 0xF32CFC5B538AFA89ULL, 0xB5E71911D44501FBULL, 0x47AB1EFF0A417FF6ULL, 0x06D89F71CAB8351FULL },
    { 0x71930c11d782e155ULL, 0xa6bb947cffbe3323ULL, 0xaa303344d4741444ULL, 0x2c3b3f0d26594943ULL },
    0x87d20782e4866389ULL
};

static const struct fparams FR = {
    { 0x43E1F593F0000001ULL, 0x2833E84879B97091ULL, 0xB85045B68181585DULL, 0x30644E72E131A029ULL },
    { 0x87c3eb27e0000002ULL, 0x5067d090f372e122ULL, 0x70a08b6d0302b0baULL, 0x60c89ce5c2634053ULL },
    { 0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL },
    { 0x1BB8E645AE216DA7ULL, 0x53FE3AB1E35C59E3ULL, 0x8C49833D53BB8085ULL, 0x0216D0B17F4E44A5ULL },
    { 0x93e7cede4a0329b3ULL, 0x7d4fdca77a96c167ULL, 0x8be4ba08b19a750aULL, 0x1cbd5653a5661c25ULL },
    0xc2e1f593efffffffULL
};

/* fr only: fr.hpp:59-79 */
static const uint64_t FR_ROOT_OF_UNITY[4] = { 0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL,
                                              0x1860ef942963f9e7ULL };
#define FR_ROOT_LOG_SIZE 28
static const uint64_t FR_GENERATOR[4] = { 0x1b0d0ef99fffffe6ULL, 0xeaba68a3a32a913fULL, 0x47d8eb76d8dd0689ULL,
                                          0x15d0085520f5bbc3ULL };
static const uint64_t FR_GENERATOR_INV[4] = { 0xd745397409999999ULL, 0xb4ada7d483c3efa8ULL, 0xc49ca2f8e57f3161ULL,
                                              0x162a3754ac156cb3ULL };
static const uint64_t FR_ALT_GENERATOR[4] = { 0x3057819e4fffffdbULL, 0x307f6d866832bb01ULL, 0x5c65ec9f484e3a89ULL,
                                              0x0180a96573d3d9f8ULL };
/* g1.hpp:14-16 */
static const uint64_t G1_ONE_Y[4] = { 0xa6ba871b8b1e1b3aULL, 0x14f1d651eb8e167bULL, 0xccdd46def0f28c58ULL,
                                      0x1c14ef83340fbe5eULL };
static const uint64_t G1_B[4] = { 0x7a17caa950ad28d7ULL, 0x1f6ac17ae15521b9ULL, 0x334bea4e696bd284ULL,
                                  0x2a1f6744ce179d8eULL };

static const struct fparams* P(int f)
{
    return f == ORC_FQ ? &FQ : &FR;
}

const uint64_t* orc_const(int f, const char* name)
{
    const struct fparams* p = P(f);
    if (!strcmp(name, "modulus")) return p->mod;
    if (!strcmp(name, "twice_modulus")) return p->twice_mod;
    if (!strcmp(name, "one")) return p->one;
    if (!strcmp(name, "r_squared")) return p->r_squared;
    if (!strcmp(name, "beta")) return p->beta;
    if (!strcmp(name, "generator")) return FR_GENERATOR;
    if (!strcmp(name, "generator_inverse")) return FR_GENERATOR_INV;
    if (!strcmp(name, "alt_generator")) return FR_ALT_GENERATOR;
    if (!strcmp(name, "root_of_unity")) return FR_ROOT_OF_UNITY;
    if (!strcmp(name, "g1_one_y")) return G1_ONE_Y;
    if (!strcmp(name, "g1_b")) return G1_B;
    return NULL;
}

/* ------------------------------------------------------------------------- */
/* limb primitives: field_impl_int128.tcc:16-38                              */
/* ------------------------------------------------------------------------- */
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t cin, uint64_t* cout)
{
    u128 s = (u128)a + b + cin;
    *cout = (uint64_t)(s >> 64);
    return (uint64_t)s;
}
/* a - (b + (borrow_in >> 63)); borrow_out is the all-ones/zero high word */
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t bin, uint64_t* bout)
{
    u128 d = (u128)a - ((u128)b + (bin >> 63));
    *bout = (uint64_t)(d >> 64);
    return (uint64_t)d;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t cin, uint64_t* cout)
{
    u128 s = (u128)a + (u128)b * c + cin;
    *cout = (uint64_t)(s >> 64);
    return (uint64_t)s;
}

/* r = a - b, then + (m & borrow): field_impl_int128.tcc:40-70 (m = modulus or twice_modulus) */
static void sub_then_fix(const uint64_t a[4], const uint64_t b[4], const uint64_t m[4], uint64_t r[4])
{
    uint64_t bw = 0, cy = 0, t[4];
    for (int i = 0; i < 4; i++) t[i] = sbb(a[i], b[i], bw, &bw);
    for (int i = 0; i < 4; i++) r[i] = adc(t[i], m[i] & bw, cy, &cy);
}

/* field_impl_int128.tcc:114-137 */
void orc_mul_512(const uint64_t a[4], const uint64_t b[4], uint64_t r[8])
{
    uint64_t t[8] = { 0 };
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) t[i + j] = mac(t[i + j], a[i], b[j], c, &c);
        t[i + 4] = c;
    }
    memcpy(r, t, sizeof(t));
}

/* field_impl_int128.tcc:72-110 */
static void mont_reduce(const struct fparams* p, uint64_t t[8], uint64_t out[4])
{
    uint64_t c2 = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t k = t[i] * p->r_inv, c = 0, sink;
        sink = mac(t[i], k, p->mod[0], 0, &c);
        (void)sink;
        for (int j = 1; j < 4; j++) t[i + j] = mac(t[i + j], k, p->mod[j], c, &c);
        t[i + 4] = adc(t[i + 4], c2, c, &c2);
    }
    for (int i = 0; i < 4; i++) out[i] = t[4 + i];
}

void orc_mul_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    uint64_t t[8];
    orc_mul_512(a, b, t);
    mont_reduce(P(f), t, r);
}
void orc_sqr_coarse(int f, const uint64_t a[4], uint64_t r[4])
{
    orc_mul_coarse(f, a, a, r);
}
/* field_impl_int128.tcc:248-255: product, reduce, one conditional subtraction */
void orc_mul(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    uint64_t t[4];
    orc_mul_coarse(f, a, b, t);
    sub_then_fix(t, P(f)->mod, P(f)->mod, r);
}
void orc_sqr(int f, const uint64_t a[4], uint64_t r[4])
{
    orc_mul(f, a, a, r);
}
/* field_impl_int128.tcc:169-177 */
void orc_add_noreduce(const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) r[i] = adc(a[i], b[i], c, &c);
}
/* :155-160 */
void orc_add(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    uint64_t t[4];
    orc_add_noreduce(a, b, t);
    sub_then_fix(t, P(f)->mod, P(f)->mod, r);
}
/* :162-167 */
void orc_add_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    uint64_t t[4];
    orc_add_noreduce(a, b, t);
    sub_then_fix(t, P(f)->twice_mod, P(f)->twice_mod, r);
}
/* :205-215 */
void orc_sub(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    sub_then_fix(a, b, P(f)->mod, r);
}
void orc_sub_coarse(int f, const uint64_t a[4], const uint64_t b[4], uint64_t r[4])
{
    sub_then_fix(a, b, P(f)->twice_mod, r);
}
/* :148-152 */
void orc_reduce_once(int f, const uint64_t a[4], uint64_t r[4])
{
    sub_then_fix(a, P(f)->mod, P(f)->mod, r);
}
/* field.hpp:123-126 */
void orc_neg(int f, const uint64_t a[4], uint64_t r[4])
{
    orc_sub(f, P(f)->mod, a, r);
}

static int gt4(const uint64_t a[4], const uint64_t b[4])
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return 0;
}
static int is_zero4(const uint64_t a[4])
{
    return (a[0] | a[1] | a[2] | a[3]) == 0;
}
static int eq4(const uint64_t a[4], const uint64_t b[4])
{
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

/* field.hpp:224-236: note the loop compares against modulus + 1 */
static void reduce_loop(const struct fparams* p, uint64_t r[4])
{
    uint64_t mp1[4] = { p->mod[0] + 1, p->mod[1], p->mod[2], p->mod[3] };
    while (gt4(r, mp1)) sub_then_fix(r, p->mod, p->mod, r);
}
void orc_to_mont(int f, const uint64_t a[4], uint64_t r[4])
{
    uint64_t t[4];
    memcpy(t, a, 32);
    reduce_loop(P(f), t);
    orc_mul(f, t, P(f)->r_squared, r);
}
void orc_from_mont(int f, const uint64_t a[4], uint64_t r[4])
{
    static const uint64_t one_raw[4] = { 1, 0, 0, 0 };
    orc_mul(f, a, one_raw, r);
}

/* field.hpp:258-292 */
void orc_pow(int f, const uint64_t a[4], const uint64_t e[4], uint64_t r[4])
{
    uint64_t acc[4];
    if (is_zero4(a)) {
        memset(r, 0, 32);
        return;
    }
    memcpy(acc, a, 32);
    int i = 255;
    while (!((e[i >> 6] >> (i & 63)) & 1)) --i;
    for (--i; i >= 0; --i) {
        orc_sqr(f, acc, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) orc_mul(f, acc, a, acc);
    }
    reduce_loop(P(f), acc);
    memcpy(r, acc, 32);
}
/* field.hpp:294-339 */
void orc_pow_small(int f, const uint64_t a[4], uint64_t e, uint64_t r[4])
{
    if (e == 0) {
        memcpy(r, P(f)->one, 32);
        return;
    }
    if (e == 1) {
        memcpy(r, a, 32);
        return;
    }
    if (e == 2) {
        orc_sqr(f, a, r);
        return;
    }
    uint64_t ev[4] = { e, 0, 0, 0 }, base[4];
    memcpy(base, a, 32);
    /* same square-and-multiply chain as pow() restricted to 64 exponent bits; unlike pow() there is no
     * early-out for a == 0, but 0^e = 0 falls out of the chain */
    uint64_t acc[4];
    memcpy(acc, base, 32);
    int i = 63;
    while (!((ev[0] >> i) & 1)) --i;
    for (--i; i >= 0; --i) {
        orc_sqr(f, acc, acc);
        if ((ev[0] >> i) & 1) orc_mul(f, acc, base, acc);
    }
    reduce_loop(P(f), acc);
    memcpy(r, acc, 32);
}
/* field.hpp:343-346 */
void orc_invert(int f, const uint64_t a[4], uint64_t r[4])
{
    const struct fparams* p = P(f);
    uint64_t e[4] = { p->mod[0] - 2, p->mod[1], p->mod[2], p->mod[3] };
    orc_pow(f, a, e, r);
}
/* field.hpp:503-522 */
void orc_batch_invert(int f, uint64_t* coeffs, size_t n)
{
    uint64_t* tmp = (uint64_t*)malloc(32 * (n ? n : 1));
    uint64_t acc[4], t0[4];
    memcpy(acc, P(f)->one, 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(tmp + 4 * i, acc, 32);
        orc_mul(f, acc, coeffs + 4 * i, acc);
    }
    orc_invert(f, acc, acc);
    for (size_t i = n; i-- > 0;) {
        orc_mul(f, acc, tmp + 4 * i, t0);
        orc_mul(f, acc, coeffs + 4 * i, acc);
        memcpy(coeffs + 4 * i, t0, 32);
    }
    free(tmp);
}
/* field.hpp:487-494 */
void orc_get_root_of_unity(size_t degree_log2, uint64_t r[4])
{
    memcpy(r, FR_ROOT_OF_UNITY, 32);
    for (size_t i = FR_ROOT_LOG_SIZE; i > degree_log2; --i) orc_sqr(ORC_FR, r, r);
}

/* ------------------------------------------------------------------------- */
/* endomorphism split: field.hpp:413-485                                     */
/* ------------------------------------------------------------------------- */
void orc_split_endo(const uint64_t k[4], uint64_t k1[2], uint64_t k2[2])
{
    static const uint64_t g1[4] = { 0x7a7bd9d4391eb18dULL, 0x4ccef014a773d2cfULL, 0x2ULL, 0 };
    static const uint64_t g2[4] = { 0xd91d232ec7e0b3d7ULL, 0x2ULL, 0, 0 };
    static const uint64_t minus_b1[4] = { 0x8211bbeb7d4f1128ULL, 0x6f4d8248eeb859fcULL, 0, 0 };
    static const uint64_t b2[4] = { 0x89d3256894d213e3ULL, 0, 0, 0 };
    uint64_t c1[8], c2[8], q1[8], q2[8], t1[4], t2[4];
    orc_mul_512(g2, k, c1); /* c1 = (g2 * k) >> 256 : take high half */
    orc_mul_512(g1, k, c2);
    orc_mul_512(c1 + 4, minus_b1, q1);
    orc_mul_512(c2 + 4, b2, q2);
    orc_sub(ORC_FR, q2, q1, t1);       /* low 256 bits each, modular subtract */
    orc_mul(ORC_FR, t1, FR.beta, t2);  /* t1 * lambda (lambda stored in Montgomery form => plain product) */
    orc_add(ORC_FR, k, t2, t2);
    k2[0] = t1[0];
    k2[1] = t1[1];
    k1[0] = t2[0];
    k1[1] = t2[1];
}

/* ------------------------------------------------------------------------- */
/* wNAF: groups/wnaf.hpp:15-55                                               */
/* ------------------------------------------------------------------------- */
uint32_t orc_get_wnaf_bits(const uint64_t* scalar, size_t bits, size_t bit_position)
{
    size_t lo_idx = bit_position >> 6;
    size_t hi_idx = (bit_position + bits - 1) >> 6;
    size_t sh = bit_position & 63;
    uint32_t lo = (uint32_t)(scalar[lo_idx] >> sh);
    uint32_t hi = 0;
    if (hi_idx != lo_idx) hi = (uint32_t)(scalar[hi_idx] << (64 - sh));
    return (lo | hi) & ((1U << (uint32_t)bits) - 1U);
}

void orc_fixed_wnaf(const uint64_t scalar[2], uint32_t* wnaf, uint8_t* skew, size_t stride, size_t wnaf_bits)
{
    const size_t entries = (127 + wnaf_bits - 1) / wnaf_bits;
    const uint32_t w = (uint32_t)wnaf_bits;
    *skew = (uint8_t)((scalar[0] & 1) == 0);
    uint32_t prev = orc_get_wnaf_bits(scalar, wnaf_bits, 0) + *skew;
    for (size_t i = 1; i + 1 < entries; ++i) {
        uint32_t slice = orc_get_wnaf_bits(scalar, wnaf_bits, i * wnaf_bits);
        uint32_t even = ((slice & 1U) == 0U);
        wnaf[(entries - i) * stride] = (((prev - (even << w)) ^ (0U - even)) >> 1U) | (even << 31U);
        prev = slice + even;
    }
    size_t final_bits = 127 - (127 / wnaf_bits) * wnaf_bits;
    uint32_t slice = orc_get_wnaf_bits(scalar, final_bits, (entries - 1) * wnaf_bits);
    uint32_t even = ((slice & 1U) == 0U);
    wnaf[stride] = (((prev - (even << w)) ^ (0U - even)) >> 1U) | (even << 31U);
    wnaf[0] = ((slice + even) >> 1U);
}

/* ------------------------------------------------------------------------- */
/* G1: groups/group.hpp                                                      */
/* ------------------------------------------------------------------------- */
#define X(p) (p)
#define Y(p) ((p) + 4)
#define Z(p) ((p) + 8)
#define MSB_SET(y) (((y)[3] >> 63) == 1)

void orc_g1_set_infinity(uint64_t p[12])
{
    Y(p)[3] = 1ULL << 63; /* field.hpp:199-202: overwrites the limb */
}
int orc_g1_is_infinity(const uint64_t* p)
{
    return MSB_SET(Y(p));
}
void orc_g1_one_affine(uint64_t r[8])
{
    memcpy(X(r), FQ.one, 32);
    memcpy(Y(r), G1_ONE_Y, 32);
}
void orc_g1_neg_affine(const uint64_t p[8], uint64_t r[8])
{
    memcpy(X(r), X(p), 32);
    orc_neg(ORC_FQ, Y(p), Y(r));
}

#define FQM(a, b, r) orc_mul(ORC_FQ, a, b, r)
#define FQMC(a, b, r) orc_mul_coarse(ORC_FQ, a, b, r)
#define FQSC(a, r) orc_sqr_coarse(ORC_FQ, a, r)
#define FQAC(a, b, r) orc_add_coarse(ORC_FQ, a, b, r)
#define FQSUBC(a, b, r) orc_sub_coarse(ORC_FQ, a, b, r)
#define FQRED(a, r) orc_reduce_once(ORC_FQ, a, r)

/* group.hpp:153-217 */
void orc_g1_dbl(const uint64_t p1[12], uint64_t out[12])
{
    if (MSB_SET(Y(p1))) {
        orc_g1_set_infinity(out);
        return;
    }
    uint64_t t0[4], t1[4], t2[4], t3[4], x2[4], y2[4], z2[4];
    orc_add_noreduce(Z(p1), Z(p1), z2);
    FQMC(z2, Y(p1), z2);
    FQRED(z2, z2);
    FQSC(X(p1), t0);
    FQSC(Y(p1), t1);
    FQSC(t1, t2);
    FQAC(t1, X(p1), t1);
    FQSC(t1, t1);
    FQAC(t0, t2, t3);
    FQSUBC(t1, t3, t1);
    FQAC(t1, t1, t1);
    FQAC(t0, t0, t3);
    FQAC(t3, t0, t3);
    FQAC(t1, t1, t0);
    FQSC(t3, x2);
    FQSUBC(x2, t0, x2);
    FQRED(x2, x2);
    /* 8*T2 with a coarse reduction after every doubling: field_impl_int128.tcc:187-195 */
    FQAC(t2, t2, t2);
    FQAC(t2, t2, t2);
    FQAC(t2, t2, t2);
    FQSUBC(t1, x2, y2);
    FQMC(y2, t3, y2);
    FQSUBC(y2, t2, y2);
    FQRED(y2, y2);
    memcpy(X(out), x2, 32);
    memcpy(Y(out), y2, 32);
    memcpy(Z(out), z2, 32);
}

/* group.hpp:219-322 */
void orc_g1_mixed_add(const uint64_t p1[12], const uint64_t p2[8], uint64_t out[12])
{
    if (MSB_SET(Y(p1))) {
        memcpy(X(out), X(p2), 32);
        memcpy(Y(out), Y(p2), 32);
        memcpy(Z(out), FQ.one, 32);
        return;
    }
    uint64_t t0[4], t1[4], t2[4], t3[4], x3[4], y3[4], z3[4];
    FQSC(Z(p1), t0);
    FQM(X(p2), t0, t1);
    orc_sub(ORC_FQ, t1, X(p1), t1);
    FQMC(Z(p1), t0, t2);
    FQM(t2, Y(p2), t2);
    orc_sub(ORC_FQ, t2, Y(p1), t2);
    if (is_zero4(t1)) {
        if (is_zero4(t2)) {
            orc_g1_dbl(p1, out);
        } else {
            /* the reference leaves x,z of p3 untouched; carry p1's over so in-place use matches */
            uint64_t tmp[12];
            memcpy(tmp, p1, 96);
            orc_g1_set_infinity(tmp);
            memcpy(out, tmp, 96);
        }
        return;
    }
    /* field_impl_int128.tcc:197-201 : t2 = 2*t2 ; z3 = z1 + t1 (no reductions) */
    orc_add_noreduce(t2, t2, t2);
    orc_add_noreduce(Z(p1), t1, z3);
    FQSC(t1, t3);
    FQAC(t0, t3, t0);
    FQSC(z3, z3);
    FQSUBC(z3, t0, z3);
    FQRED(z3, z3);
    /* 4*T3 coarse: field_impl_int128.tcc:179-185 */
    FQAC(t3, t3, t3);
    FQAC(t3, t3, t3);
    FQMC(t1, t3, t1);
    FQMC(t3, X(p1), t3);
    FQAC(t3, t3, t0);
    FQAC(t0, t1, t0);
    FQSC(t2, x3);
    FQSUBC(x3, t0, x3);
    FQSUBC(t3, x3, t3);
    FQRED(x3, x3);
    FQMC(t1, Y(p1), t1);
    FQAC(t1, t1, t1);
    FQMC(t3, t2, t3);
    FQSUBC(t3, t1, y3);
    FQRED(y3, y3);
    memcpy(X(out), x3, 32);
    memcpy(Y(out), y3, 32);
    memcpy(Z(out), z3, 32);
}

/* group.hpp:324-448 */
void orc_g1_add(const uint64_t p1[12], const uint64_t p2[12], uint64_t out[12])
{
    int z1 = MSB_SET(Y(p1)), z2 = MSB_SET(Y(p2));
    if (z1 || z2) {
        if (z1 && !z2) {
            memmove(out, p2, 96);
            return;
        }
        if (z2 && !z1) {
            memmove(out, p1, 96);
            return;
        }
        uint64_t tmp[12];
        memcpy(tmp, out, 96);
        orc_g1_set_infinity(tmp);
        memcpy(out, tmp, 96);
        return;
    }
    uint64_t z1z1[4], z2z2[4], u1[4], u2[4], s1[4], s2[4], F[4], H[4], I[4], J[4], x3[4], y3[4], z3[4];
    FQSC(Z(p1), z1z1);
    FQSC(Z(p2), z2z2);
    FQMC(X(p1), z2z2, u1);
    FQMC(X(p2), z1z1, u2);
    FQMC(Z(p2), z2z2, s1);
    FQMC(Z(p1), z1z1, s2);
    FQMC(s1, Y(p1), s1);
    FQMC(s2, Y(p2), s2);
    FQSUBC(u2, u1, H);
    FQRED(H, H);
    FQSUBC(s2, s1, F);
    FQRED(F, F);
    if (is_zero4(H)) {
        if (is_zero4(F)) {
            orc_g1_dbl(p1, out);
        } else {
            uint64_t tmp[12];
            memcpy(tmp, out, 96);
            orc_g1_set_infinity(tmp);
            memcpy(out, tmp, 96);
        }
        return;
    }
    orc_add_noreduce(F, F, F);
    orc_add_noreduce(H, H, I);
    FQSC(I, I);
    FQMC(H, I, J);
    FQMC(u1, I, u1);
    FQAC(u1, u1, u2);
    FQAC(u2, J, u2);
    FQSC(F, x3);
    FQSUBC(x3, u2, x3);
    FQRED(x3, x3);
    FQMC(J, s1, J);
    FQAC(J, J, J);
    FQSUBC(u1, x3, y3);
    FQMC(y3, F, y3);
    FQSUBC(y3, J, y3);
    FQRED(y3, y3);
    FQAC(Z(p1), Z(p2), z3);
    FQAC(z1z1, z2z2, z1z1);
    FQSC(z3, z3);
    FQSUBC(z3, z1z1, z3);
    FQM(z3, H, z3);
    memcpy(X(out), x3, 32);
    memcpy(Y(out), y3, 32);
    memcpy(Z(out), z3, 32);
}

/* group.hpp:450-468 */
void orc_g1_normalize(const uint64_t p[12], uint64_t r[12])
{
    uint64_t zi[4], zz[4], zzz[4], out[12];
    int inf = MSB_SET(Y(p));
    orc_invert(ORC_FQ, Z(p), zi);
    orc_sqr(ORC_FQ, zi, zz);
    FQM(zi, zz, zzz);
    FQM(X(p), zz, X(out));
    FQM(Y(p), zzz, Y(out));
    memcpy(Z(out), FQ.one, 32);
    if (inf) orc_g1_set_infinity(out);
    memcpy(r, out, 96);
}

/* group.hpp:474-534 */
void orc_g1_batch_normalize(uint64_t* pts, size_t n)
{
    uint64_t* tmp = (uint64_t*)malloc(32 * (n ? n : 1));
    uint64_t acc[4], zi[4], zz[4], zzz[4];
    memcpy(acc, FQ.one, 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(tmp + 4 * i, acc, 32);
        if (!MSB_SET(Y(pts + 12 * i))) FQM(acc, Z(pts + 12 * i), acc);
    }
    orc_invert(ORC_FQ, acc, acc);
    for (size_t i = n; i-- > 0;) {
        uint64_t* q = pts + 12 * i;
        if (!MSB_SET(Y(q))) {
            FQM(acc, tmp + 4 * i, zi);
            orc_sqr(ORC_FQ, zi, zz);
            FQM(zi, zz, zzz);
            FQM(X(q), zz, X(q));
            FQM(Y(q), zzz, Y(q));
            FQM(acc, Z(q), acc);
        }
        memcpy(Z(q), FQ.one, 32);
    }
    free(tmp);
}

/* group.hpp:536-550 */
int orc_g1_on_curve_affine(const uint64_t p[8])
{
    if (MSB_SET(Y(p))) return 0;
    uint64_t xxx[4], yy[4];
    orc_sqr(ORC_FQ, X(p), xxx);
    FQM(X(p), xxx, xxx);
    orc_add(ORC_FQ, xxx, G1_B, xxx);
    orc_sqr(ORC_FQ, Y(p), yy);
    orc_from_mont(ORC_FQ, xxx, xxx);
    orc_from_mont(ORC_FQ, yy, yy);
    return eq4(xxx, yy);
}

/* Plain double-and-add over the canonical scalar.  The reference's windowed endomorphism ladder
 * (group.hpp:653-760) computes the same group element; after normalisation the two are limb-identical,
 * which is how the reference's own tests compare (test_scalar_multiplication.cpp:94-103). */
void orc_g1_scalar_mul(const uint64_t p[8], const uint64_t scalar_mont[4], uint64_t r[12])
{
    uint64_t k[4], acc[12];
    orc_from_mont(ORC_FR, scalar_mont, k);
    memset(acc, 0, sizeof(acc));
    orc_g1_set_infinity(acc);
    for (int i = 255; i >= 0; --i) {
        orc_g1_dbl(acc, acc);
        if ((k[i >> 6] >> (i & 63)) & 1) orc_g1_mixed_add(acc, p, acc);
    }
    if (orc_g1_is_infinity(acc)) {
        memset(r, 0, 96);
        orc_g1_set_infinity(r);
        return;
    }
    orc_g1_batch_normalize(acc, 1);
    memcpy(r, acc, 96);
}

/* ------------------------------------------------------------------------- */
/* MSM: curves/bn254/scalar_multiplication.cpp                               */
/* ------------------------------------------------------------------------- */
/* :21-81 (the >=144834 branch is unreachable behind >=100000, kept out) */
size_t orc_get_optimal_bucket_width(size_t n)
{
    static const struct {
        size_t min_points, width;
    } tab[] = { { 14617149, 21 }, { 2139094, 18 }, { 100000, 15 }, { 25067, 12 }, { 13926, 11 }, { 7659, 10 },
                { 2436, 9 },      { 376, 7 },      { 231, 6 },     { 97, 5 },     { 35, 4 },     { 10, 3 },
                { 2, 2 } };
    for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++) {
        if (n >= tab[i].min_points) return tab[i].width;
    }
    return 1;
}

/* :131-140 -- back to front so that table may alias points */
void orc_generate_point_table(const uint64_t* points, uint64_t* table, size_t n)
{
    for (size_t i = n; i-- > 0;) {
        uint64_t x[4], y[4];
        memcpy(x, points + 8 * i, 32);
        memcpy(y, points + 8 * i + 4, 32);
        memcpy(table + 16 * i, x, 32);
        memcpy(table + 16 * i + 4, y, 32);
        FQM(x, FQ.beta, table + 16 * i + 8);
        orc_neg(ORC_FQ, y, table + 16 * i + 12);
    }
}

/* group_impl_int128.tcc / group_impl_asm.tcc:71-153: dest = predicate ? (x, p - y) : src */
static void cond_negate_affine(const uint64_t* src, uint64_t* dst, uint64_t predicate)
{
    memcpy(dst, src, 64);
    if (predicate) orc_neg(ORC_FQ, Y(src), Y(dst));
}

/* :457-476 + :576-648 + :265-308 */
void orc_pippenger(const uint64_t* scalars_mont, const uint64_t* table, size_t n, size_t forced_bucket_width,
                   uint64_t out[12])
{
    if (n == 0) {
        memcpy(X(out), FQ.one, 32);
        memcpy(Y(out), G1_ONE_Y, 32);
        memcpy(Z(out), FQ.one, 32);
        orc_g1_set_infinity(out);
        return;
    }
    const size_t c = forced_bucket_width ? forced_bucket_width : orc_get_optimal_bucket_width(n);
    const size_t num_points = 2 * n;
    const size_t wbits = c + 1;
    const size_t rounds = (127 + wbits - 1) / wbits;
    const size_t num_buckets = (size_t)1 << c;

    uint64_t* buckets = (uint64_t*)malloc(96 * num_buckets);
    uint32_t* wnaf = (uint32_t*)calloc(rounds * num_points + 1, sizeof(uint32_t));
    uint8_t* skew = (uint8_t*)calloc(num_points + 1, 1);
    memset(buckets, 0, 96 * num_buckets);
    for (size_t b = 0; b < num_buckets; b++) orc_g1_set_infinity(buckets + 12 * b);

    for (size_t i = 0; i < n; i++) {
        uint64_t k[4], k1[2], k2[2];
        orc_from_mont(ORC_FR, scalars_mont + 4 * i, k);
        orc_split_endo(k, k1, k2);
        orc_fixed_wnaf(k1, wnaf + 2 * i, skew + 2 * i, num_points, wbits);
        orc_fixed_wnaf(k2, wnaf + 2 * i + 1, skew + 2 * i + 1, num_points, wbits);
    }

    uint64_t acc[12], running[12], tmp_pt[8];
    memset(acc, 0, sizeof(acc));
    memset(running, 0, sizeof(running));
    orc_g1_set_infinity(acc);
    for (size_t r = 0; r < rounds; r++) {
        if (r == rounds - 1) {
            for (size_t j = 0; j < num_points; j++) {
                if (skew[j]) {
                    orc_g1_neg_affine(table + 8 * j, tmp_pt);
                    orc_g1_mixed_add(buckets, tmp_pt, buckets);
                }
            }
        }
        const uint32_t* row = wnaf + r * num_points;
        for (size_t j = 0; j < num_points; j++) {
            uint32_t e = row[j];
            uint64_t* bk = buckets + 12 * (size_t)(e & 0x0fffffffU);
            cond_negate_affine(table + 8 * j, tmp_pt, (e >> 31) & 1);
            orc_g1_mixed_add(bk, tmp_pt, bk);
        }
        if (r > 0) {
            for (size_t j = 0; j < c; j++) orc_g1_dbl(acc, acc);
        }
        orc_g1_set_infinity(running);
        for (size_t b = num_buckets - 1; b > 0; b--) {
            orc_g1_add(running, buckets + 12 * b, running);
            orc_g1_add(acc, running, acc);
            orc_g1_set_infinity(buckets + 12 * b);
        }
        orc_g1_add(running, buckets, running);
        orc_g1_dbl(acc, acc);
        orc_g1_add(acc, running, acc);
        orc_g1_set_infinity(buckets);
    }
    memcpy(out, acc, 96);
    free(buckets);
    free(wnaf);
    free(skew);
}

/* :650-772 with `threads` standing in for omp_get_max_threads() */
int orc_batched_msm(struct orc_msm_job* jobs, size_t num_jobs, size_t threads)
{
    if (num_jobs == 0) return 0;
    size_t n = jobs[0].num_elements;
    for (size_t i = 1; i < num_jobs; i++) {
        if (jobs[i].num_elements != n) return 1; /* reference prints and returns with outputs untouched */
    }
    if (threads == 0) threads = 1;
    size_t per_job = threads / num_jobs;
    if (per_job * num_jobs != threads) ++per_job;
    size_t each = n / per_job;
    size_t rem = n > each * per_job ? n - each * per_job : 0;
    uint64_t* outs = (uint64_t*)malloc(96 * num_jobs);
    for (size_t i = 0; i < num_jobs; i++) {
        size_t off = 0;
        uint64_t sum[12], part[12];
        for (size_t j = 0; j < per_job; j++) {
            size_t len = each + (j == 0 ? rem : 0);
            orc_pippenger(jobs[i].scalars + 4 * off, jobs[i].points + 16 * off, len, 0, part);
            if (j == 0) memcpy(sum, part, 96);
            else orc_g1_add(sum, part, sum);
            off += len;
        }
        memcpy(outs + 12 * i, sum, 96);
    }
    orc_g1_batch_normalize(outs, num_jobs);
    for (size_t i = 0; i < num_jobs; i++) memcpy(jobs[i].output, outs + 12 * i, 96);
    free(outs);
    return 0;
}

/* Synthetic SRS x^i.G (shape of test/test_preprocess.cpp:24-35 and io.hpp:176-178).
 * Fixed-base 8-bit window table over the canonical scalar x^i, batch-normalised at the end. */
void orc_make_srs(const uint64_t x_mont[4], size_t n, uint64_t* out)
{
    enum { WIN = 8, NWIN = 32, TSZ = 256 };
    uint64_t* tab = (uint64_t*)malloc((size_t)96 * NWIN * TSZ); /* tab[w][d] = d * 2^(8w) * G, Jacobian then affine */
    uint64_t base[12];
    orc_g1_one_affine(base);
    memcpy(Z(base), FQ.one, 32);
    for (int w = 0; w < NWIN; w++) {
        uint64_t* row = tab + (size_t)12 * TSZ * w;
        memset(row, 0, 96);
        orc_g1_set_infinity(row);
        memcpy(row + 12, base, 96);
        for (int d = 2; d < TSZ; d++) orc_g1_add(row + 12 * (d - 1), base, row + 12 * d);
        for (int k = 0; k < WIN; k++) orc_g1_dbl(base, base);
    }
    orc_g1_batch_normalize(tab, (size_t)NWIN * TSZ);
    uint64_t* jac = (uint64_t*)malloc(96 * (n ? n : 1));
    uint64_t s[4];
    memcpy(s, FR.one, 32);
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4], acc[12];
        orc_from_mont(ORC_FR, s, k);
        memset(acc, 0, 96);
        orc_g1_set_infinity(acc);
        for (int w = 0; w < NWIN; w++) {
            unsigned d = (unsigned)((k[w >> 3] >> ((w & 7) * 8)) & 0xff);
            if (d) orc_g1_mixed_add(acc, tab + (size_t)12 * (TSZ * w + d), acc);
        }
        memcpy(jac + 12 * i, acc, 96);
        orc_mul(ORC_FR, s, x_mont, s);
    }
    orc_g1_batch_normalize(jac, n);
    for (size_t i = 0; i < n; i++) memcpy(out + 8 * i, jac + 12 * i, 64);
    free(jac);
    free(tab);
}

/* ------------------------------------------------------------------------- */
/* NTT: polynomials/polynomial_arithmetic.cpp, evaluation_domain.cpp         */
/* ------------------------------------------------------------------------- */
static uint32_t reverse_bits(uint32_t x, uint32_t bit_length) /* :14-21 */
{
    uint32_t r = 0;
    for (uint32_t i = 0; i < bit_length; i++) r |= ((x >> i) & 1U) << (bit_length - 1 - i);
    return r;
}

static size_t log2_exact(size_t n)
{
    size_t l = 0;
    while (((size_t)1 << l) < n) ++l;
    return l;
}

/* evaluation_domain.cpp:33-54: round s (m = 2^(s+1)) holds round_root^j, j < m, built with coarse products */
static void build_round_roots(const uint64_t root[4], size_t n, uint64_t* table, uint64_t** rounds)
{
    size_t num_rounds = log2_exact(n);
    if (num_rounds < 2) return;
    rounds[0] = table;
    for (size_t i = 1; i + 1 < num_rounds; i++) rounds[i] = rounds[i - 1] + 4 * ((size_t)1 << i);
    for (size_t i = 0; i + 1 < num_rounds; i++) {
        size_t m = (size_t)1 << (i + 1);
        uint64_t rr[4];
        orc_pow_small(ORC_FR, root, n / (2 * m), rr);
        memcpy(rounds[i], FR.one, 32);
        for (size_t j = 1; j < m; j++) orc_mul_coarse(ORC_FR, rounds[i] + 4 * (j - 1), rr, rounds[i] + 4 * j);
    }
}

/* polynomial_arithmetic.cpp:129-264 (same arithmetic, one thread) */
static void fft_inner(uint64_t* coeffs, size_t n, const uint64_t root[4])
{
    size_t lg = log2_exact(n);
    uint64_t* scratch = (uint64_t*)malloc(32 * n);
    uint64_t* table = (uint64_t*)malloc(32 * (n > 1 ? n : 2));
    uint64_t* rounds[64] = { 0 };
    build_round_roots(root, n, table, rounds);
    for (size_t i = 0; i < n; i++) memcpy(scratch + 4 * i, coeffs + 4 * reverse_bits((uint32_t)i, (uint32_t)lg), 32);
    for (size_t i = 0; i + 1 < n; i += 2) {
        uint64_t t[4];
        memcpy(t, scratch + 4 * (i + 1), 32);
        orc_sub_coarse(ORC_FR, scratch + 4 * i, scratch + 4 * (i + 1), scratch + 4 * (i + 1));
        orc_add_coarse(ORC_FR, t, scratch + 4 * i, scratch + 4 * i);
    }
    if (n <= 2) {
        for (size_t i = 0; i < n; i++) orc_reduce_once(ORC_FR, scratch + 4 * i, coeffs + 4 * i);
    }
    for (size_t m = 2; m < n; m <<= 1) {
        const uint64_t* rr = rounds[log2_exact(m) - 1];
        int last = (m == (n >> 1));
        for (size_t i = 0; i < n / 2; i++) {
            size_t k1 = (i & ~(m - 1)) << 1, j1 = i & (m - 1);
            uint64_t* lo = scratch + 4 * (k1 + j1);
            uint64_t* hi = scratch + 4 * (k1 + j1 + m);
            uint64_t t[4];
            orc_mul_coarse(ORC_FR, rr + 4 * j1, hi, t);
            orc_sub_coarse(ORC_FR, lo, t, hi);
            orc_add_coarse(ORC_FR, lo, t, lo);
            if (last) {
                orc_reduce_once(ORC_FR, hi, coeffs + 4 * (k1 + j1 + m));
                orc_reduce_once(ORC_FR, lo, coeffs + 4 * (k1 + j1));
            }
        }
    }
    free(table);
    free(scratch);
}

/* :81-102 with one "thread": coeffs[i] *= start * shift^i */
static void scale_by_generator(uint64_t* coeffs, size_t n, const uint64_t start[4], const uint64_t shift[4])
{
    uint64_t g[4], one_shift[4];
    orc_pow_small(ORC_FR, shift, 0, one_shift);
    orc_mul_coarse(ORC_FR, start, one_shift, g);
    for (size_t i = 0; i < n; i++) {
        orc_mul(ORC_FR, coeffs + 4 * i, g, coeffs + 4 * i);
        orc_mul_coarse(ORC_FR, g, shift, g);
    }
}

int orc_ntt(uint64_t* coeffs, size_t n, int kind, const uint64_t* constant)
{
    if (n == 0 || (n & (n - 1))) return 1;
    size_t lg = log2_exact(n);
    uint64_t root[4], root_inv[4], dom[4], dom_inv[4], size_raw[4] = { n, 0, 0, 0 }, c[4];
    orc_get_root_of_unity(lg, root);                 /* evaluation_domain.cpp:64 */
    orc_invert(ORC_FR, root, root_inv);              /* :65 */
    orc_to_mont(ORC_FR, size_raw, dom);              /* :66 */
    orc_invert(ORC_FR, dom, dom_inv);                /* :67 */
    switch (kind) {
    case ORC_FFT: /* :266 */
        fft_inner(coeffs, n, root);
        break;
    case ORC_IFFT: /* :271-277 */
        fft_inner(coeffs, n, root_inv);
        for (size_t i = 0; i < n; i++) orc_mul(ORC_FR, coeffs + 4 * i, dom_inv, coeffs + 4 * i);
        break;
    case ORC_FFT_WITH_CONSTANT: /* :279-285 */
        fft_inner(coeffs, n, root);
        for (size_t i = 0; i < n; i++) orc_mul(ORC_FR, coeffs + 4 * i, constant, coeffs + 4 * i);
        break;
    case ORC_COSET_FFT: /* :287-291 */
        scale_by_generator(coeffs, n, FR.one, FR_GENERATOR);
        fft_inner(coeffs, n, root);
        break;
    case ORC_COSET_FFT_WITH_CONSTANT: /* :293-299 */
        orc_mul(ORC_FR, FR.one, constant, c);
        scale_by_generator(coeffs, n, c, FR_GENERATOR);
        fft_inner(coeffs, n, root);
        break;
    case ORC_IFFT_WITH_CONSTANT: /* :301-309 */
        fft_inner(coeffs, n, root_inv);
        orc_mul(ORC_FR, dom_inv, constant, c);
        for (size_t i = 0; i < n; i++) orc_mul(ORC_FR, coeffs + 4 * i, c, coeffs + 4 * i);
        break;
    case ORC_COSET_IFFT: /* :311-315 */
        fft_inner(coeffs, n, root_inv);
        for (size_t i = 0; i < n; i++) orc_mul(ORC_FR, coeffs + 4 * i, dom_inv, coeffs + 4 * i);
        scale_by_generator(coeffs, n, FR.one, FR_GENERATOR_INV);
        break;
    default:
        return 2;
    }
    return 0;
}

/* :337-373 -- sum coeffs[i] z^i; the reference splits into per-thread chunks, the sum is the same residue
 * and the final __add chain leaves it canonical */
void orc_evaluate(const uint64_t* coeffs, const uint64_t z[4], size_t n, uint64_t r[4])
{
    uint64_t acc[4] = { 0, 0, 0, 0 }, zp[4], t[4];
    memcpy(zp, FR.one, 32);
    for (size_t i = 0; i < n; i++) {
        orc_mul(ORC_FR, zp, coeffs + 4 * i, t);
        orc_add(ORC_FR, acc, t, acc);
        orc_mul(ORC_FR, zp, z, zp);
    }
    memcpy(r, acc, 32);
}

/* ------------------------------------------------------------------------- */
/* deterministic inputs                                                      */
/* ------------------------------------------------------------------------- */
uint64_t orc_splitmix64(uint64_t* state)
{
    uint64_t z = (*state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

void orc_random_scalars(uint64_t seed, size_t n, uint64_t* out_mont)
{
    uint64_t st = seed;
    for (size_t i = 0; i < n; i++) {
        uint64_t raw[4];
        for (int j = 0; j < 4; j++) raw[j] = orc_splitmix64(&st);
        raw[3] &= 0x0fffffffffffffffULL;
        orc_to_mont(ORC_FR, raw, out_mont + 4 * i);
    }
}
