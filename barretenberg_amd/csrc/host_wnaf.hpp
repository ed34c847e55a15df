// host_wnaf.hpp -- the scalar preparation of the reference's CPU Pippenger, host only: the GLV split of a scalar into two 128-bit halves
// (fields/field.hpp:413-485, fr::split_into_endomorphism_scalars) and the fixed-window signed-digit ("wNAF") table entries
// (groups/wnaf.hpp:15-55).  The GPU path does not use either (signed fixed windows over the full scalar, msm.hip K0); they exist because
// scalar_multiplication::compute_wnaf_state is an extern of the translation unit the shim replaces and its OUTPUT -- the digit
// table, the skew bits, the split scalars -- is observable by whoever links it (the reference's own tests).  Product code: no oracle/.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "host_fr.hpp"

namespace bbgpu {
namespace host {

// 256 x 256 -> 512 bits, plain integers
static inline void mul_512(const uint64_t a[4], const uint64_t b[4], uint64_t r[8])
{
    uint64_t t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[i] * b[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 8; i++) r[i] = t[i];
}

// k (plain integer < 2^256) -> k1 (limbs 0, 1), k2 (limbs 0, 1) with k = k1 - lambda k2 (mod r), both below 2^128.
// Constants field.hpp:420-426; c1 = (g2 k) >> 256, c2 = (g1 k) >> 256, t1 = c2 b2 - c1 (-b1) on the low 256 bits with the field's one-step
// modular subtraction, k2 = t1, k1 = k + t1 lambda (lambda is held in Montgomery form, so the Montgomery product IS the plain product).
static inline void split_endo(const uint64_t k[4], uint64_t k1[2], uint64_t k2[2])
{
    static const uint64_t G1[4] = { 0x7a7bd9d4391eb18dULL, 0x4ccef014a773d2cfULL, 0x2ULL, 0 };
    static const uint64_t G2[4] = { 0xd91d232ec7e0b3d7ULL, 0x2ULL, 0, 0 };
    static const uint64_t MINUS_B1[4] = { 0x8211bbeb7d4f1128ULL, 0x6f4d8248eeb859fcULL, 0, 0 };
    static const uint64_t B2[4] = { 0x89d3256894d213e3ULL, 0, 0, 0 };
    static const Fr LAMBDA = { { 0x93e7cede4a0329b3ULL, 0x7d4fdca77a96c167ULL, 0x8be4ba08b19a750aULL, 0x1cbd5653a5661c25ULL } }; // fr.hpp:54-57
    uint64_t c1[8], c2[8], q1[8], q2[8];
    mul_512(G2, k, c1);
    mul_512(G1, k, c2);
    mul_512(c1 + 4, MINUS_B1, q1);
    mul_512(c2 + 4, B2, q2);
    Fr a, b, kk;
    memcpy(a.d, q2, 32);
    memcpy(b.d, q1, 32);
    memcpy(kk.d, k, 32);
    const Fr t1 = fr_sub(a, b);
    const Fr t2 = fr_add(kk, fr_mul(t1, LAMBDA));
    k2[0] = t1.d[0];
    k2[1] = t1.d[1];
    k1[0] = t2.d[0];
    k1[1] = t2.d[1];
}

static inline uint32_t wnaf_bits_at(const uint64_t* scalar, size_t bits, size_t position)
{
    const size_t lo = position >> 6, hi = (position + bits - 1) >> 6, sh = position & 63;
    uint32_t v = (uint32_t)(scalar[lo] >> sh);
    if (hi != lo) v |= (uint32_t)(scalar[hi] << (64 - sh));
    return v & ((1u << (uint32_t)bits) - 1u);
}
constexpr size_t WNAF_SCALAR_BITS = 127; // wnaf.hpp:11
static inline size_t wnaf_size(size_t bits) { return (WNAF_SCALAR_BITS + bits - 1) / bits; } // WNAF_SIZE, wnaf.hpp:13

// 127-bit scalar -> wnaf_size(w) odd signed digits of w bits, most significant first at wnaf[0], consecutive digits `stride` entries apart;
// an entry is (|d| - 1) / 2 with the sign in bit 31; the scalar is made odd first (skew = 1 if it was even: the caller subtracts the point once).
static inline void fixed_wnaf(const uint64_t scalar[2], uint32_t* wnaf, bool& skew, size_t stride, size_t w)
{
    const size_t entries = wnaf_size(w);
    skew = (scalar[0] & 1) == 0;
    uint32_t prev = wnaf_bits_at(scalar, w, 0) + (skew ? 1u : 0u);
    auto entry = [&](uint32_t carried, uint32_t next_even) { // the digit below a window whose value is even borrows 2^w from it
        return (((carried - (next_even << (uint32_t)w)) ^ (0u - next_even)) >> 1) | (next_even << 31);
    };
    for (size_t i = 1; i + 1 < entries; ++i) {
        const uint32_t slice = wnaf_bits_at(scalar, w, i * w), even = (slice & 1u) ^ 1u;
        wnaf[(entries - i) * stride] = entry(prev, even);
        prev = slice + even;
    }
    const size_t final_bits = WNAF_SCALAR_BITS - (WNAF_SCALAR_BITS / w) * w;
    const uint32_t slice = wnaf_bits_at(scalar, final_bits, (entries - 1) * w), even = (slice & 1u) ^ 1u;
    wnaf[stride] = entry(prev, even);
    wnaf[0] = (slice + even) >> 1;
}

} // namespace host
} // namespace bbgpu
