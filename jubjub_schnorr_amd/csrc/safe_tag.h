// SAFE sponge tag for a transcript of any length, computed at call time (on the device: pass 0 of the multisig kernels):
//   tag(n) = BLAKE2b-512( be32(0x80000000 | n) || be32(1) || be64(0) ) as a little-endian integer mod q
// (dusk-safe's `tag_input`: one absorb of n elements, one squeeze of 1, domain separator 0; restated from the published
// algorithm as the other constants are, tools/gen_constants.py sponge_tag -- the generated tables JJS_SPONGE_TAG /
// JJS_SPONGE_TAG_LONG hold the same values for n <= 16 / n <= 1027 and tests/test_hostbuild.py checks this code against
// them).  Used by the multisignature entry point for transcripts of more than JJS_MSIG_MAX_PARTICIPANTS participants
// (reference src/multisig.rs:326-338 takes any non-empty transcript).  Plain integer code, host and device.
#pragma once
#include <cstdint>
#include <cstring>
#include "fq29.h"

namespace jjs {

JJS_HD uint64_t b2_rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// BLAKE2b, unkeyed, 64-byte digest, message of at most 128 bytes (one block): RFC 7693
JJS_HD void blake2b_512_short(const uint8_t* msg, size_t len, uint8_t out[64]) {
    constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                   0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    constexpr uint8_t SIGMA[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t h[8], m[16], v[16];
    for (int i = 0; i < 8; ++i) h[i] = IV[i];
    h[0] ^= 0x01010000ull ^ 64ull;                    // digest length 64, no key, fanout = depth = 1
    uint8_t block[128] = {0};
    for (size_t i = 0; i < len && i < 128; ++i) block[i] = msg[i];
    for (int i = 0; i < 16; ++i) {
        m[i] = 0;
        for (int b = 0; b < 8; ++b) m[i] |= (uint64_t)block[8 * i + b] << (8 * b);
    }
    for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[8 + i] = IV[i]; }
    v[12] ^= (uint64_t)len;                           // byte counter (low word); the message is one block
    v[14] = ~v[14];                                   // last block
    auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
        v[a] = v[a] + v[b] + x; v[d] = b2_rotr(v[d] ^ v[a], 32);
        v[c] = v[c] + v[d];     v[b] = b2_rotr(v[b] ^ v[c], 24);
        v[a] = v[a] + v[b] + y; v[d] = b2_rotr(v[d] ^ v[a], 16);
        v[c] = v[c] + v[d];     v[b] = b2_rotr(v[b] ^ v[c], 63);
    };
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
    for (int i = 0; i < 8; ++i) {
        h[i] ^= v[i] ^ v[8 + i];
        for (int b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
    }
}

// x = 2 x + bit (mod q) on nine 32-bit words (q < 2^255, so 2 x + 1 < 2^257 fits)
JJS_HD void safe_tag_double_add(uint32_t (&x)[9], uint32_t bit, const uint32_t (&q)[9]) {
    uint32_t carry = bit;
    for (int i = 0; i < 9; ++i) { const uint32_t hi = x[i] >> 31; x[i] = (x[i] << 1) | carry; carry = hi; }
    uint32_t d[9], borrow = 0;
    for (int i = 0; i < 9; ++i) {
        const uint64_t t = (uint64_t)x[i] - q[i] - borrow;
        d[i] = (uint32_t)t; borrow = (uint32_t)(t >> 63);
    }
    if (!borrow) for (int i = 0; i < 9; ++i) x[i] = d[i];
}

// The tag of an n-element transcript as the device wants it: Montgomery form (times 2^261 mod q), nine 29-bit limbs.
// q_words: the modulus as eight 32-bit words (JJS_Q_WORDS).
JJS_HD void safe_tag_limbs(uint32_t n_inputs, const uint32_t q_words[8], uint32_t out[9]) {
    uint8_t msg[16] = {0}, dig[64];
    const uint32_t a = 0x80000000u | n_inputs;
    msg[0] = (uint8_t)(a >> 24); msg[1] = (uint8_t)(a >> 16); msg[2] = (uint8_t)(a >> 8); msg[3] = (uint8_t)a;
    msg[7] = 1;                                       // one squeezed element; the domain separator (8 bytes) is zero
    blake2b_512_short(msg, 16, dig);
    uint32_t q[9], x[9] = {0};
    for (int i = 0; i < 8; ++i) q[i] = q_words[i];
    q[8] = 0;
    for (int bit = 511; bit >= 0; --bit) safe_tag_double_add(x, (dig[bit >> 3] >> (bit & 7)) & 1u, q);   // digest (little endian) mod q
    for (int k = 0; k < 261; ++k) safe_tag_double_add(x, 0, q);                                          // times 2^261
    for (int i = 0; i < 9; ++i) {
        const int b = 29 * i, w = b >> 5, s = b & 31;
        uint64_t v = x[w] >> s;
        if (s + 29 > 32 && w + 1 < 9) v |= (uint64_t)x[w + 1] << (32 - s);
        out[i] = (uint32_t)v & 0x1fffffffu;
    }
}

}  // namespace jjs
