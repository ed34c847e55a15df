// keccak.hpp -- Keccak-256 (original Keccak padding 0x01..0x80, rate 136, as Ethereum uses it) for the Fiat-Shamir
// challenges of the PLONK prover.  The reference hashes through ethash's keccak (src/barretenberg/keccak/keccak.c:112-134,
// hash_field_elements: every 64-bit limb big-endian, limb 0 first) and reads the digest back as four native 64-bit words
// (challenge.hpp:64-76).  Restated from the Keccak specification (FIPS 202 permutation); host only, a few hundred bytes per call.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace bbgpu {
namespace host {

static inline uint64_t rotl64(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static inline void keccak_f1600(uint64_t A[25])
{
    static const uint64_t RC[24] = { 0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
                                     0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
                                     0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
                                     0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                     0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
    static const int ROT[25] = { 0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14 }; // index x + 5y
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(A[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        A[0] ^= RC[round];
    }
}

// digest as the four little-endian 64-bit words of the 32 output bytes
static inline void keccak256(const uint8_t* data, size_t len, uint64_t out[4])
{
    const size_t rate = 136;
    uint64_t A[25];
    memset(A, 0, sizeof A);
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | data[8 * i + b];
            A[i] ^= w;
        }
        keccak_f1600(A);
        data += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    if (len) memcpy(block, data, len);
    block[len] ^= 0x01;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) {
        uint64_t w = 0;
        for (int b = 7; b >= 0; b--) w = (w << 8) | block[8 * i + b];
        A[i] ^= w;
    }
    keccak_f1600(A);
    for (int i = 0; i < 4; i++) out[i] = A[i];
}

// keccak.c:112-134: `count` field elements of 4 limbs, every limb written most-significant byte first
static inline void hash_field_elements(const uint64_t* limbs, size_t count, uint64_t out[4])
{
    uint8_t stack_buf[32 * 32]; // the prover's transcripts hold at most 22 elements
    uint8_t* buf = count <= 32 ? stack_buf : (uint8_t*)malloc(count * 32);
    for (size_t i = 0; i < count * 4; i++)
        for (int b = 0; b < 8; b++) buf[8 * i + b] = (uint8_t)(limbs[i] >> (56 - 8 * b));
    keccak256(buf, count * 32, out);
    if (buf != stack_buf) free(buf);
}

} // namespace host
} // namespace bbgpu
