// Wire formats: `JubJubAffine::to_bytes / from_bytes` (dusk-jubjub 0.15) as used by
//   Signature::from_bytes        /root/reference/src/signatures.rs:112-119   (u || R, 64 bytes)
//   PublicKey::from_bytes        src/keys/public.rs:87-93                    (32 bytes)
//   SignatureDouble::from_bytes  src/signatures/double.rs:138-148            (u || R || R', 96 bytes)
//   PublicKeyDouble::from_bytes  src/keys/public/double.rs:181-185           (pk || pk', 64 bytes)
//   PublicKeyVarGen::from_bytes  src/keys/public/var_gen.rs:68-78            (pk || generator, 64 bytes)
//   SignatureVarGen::from_bytes  src/signatures/var_gen.rs:105-112           (u || R, 64 bytes)
// A compressed point is the 32-byte little-endian v with the parity of u in bit 255.  Decoding:
// v < q, u^2 = (v^2 - 1) / (1 + d v^2), square root, sign fix.  Two corner cases are pinned by no
// reference test; this implementation rejects both, like current upstream jubjub
// (ZIP 216): v >= q, and u = 0 with the sign bit set.
//
// The square root avoids the field inversion: with y = N*D (N = v^2 - 1, D = 1 + d v^2) one
// Tonelli-Shanks pass yields r = 1/sqrt(y) and u = N*r.  q - 1 = 2^32 * t, so the pass is one public
// exponentiation y^((t-1)/2) plus the 2-adic correction, done a byte at a time with table lookups
// (48 squarings and ~14 products instead of the ~500 squarings of the textbook loop).
#pragma once
#include "ed29.h"

namespace jjs {

// Tables for the 2-adic part of the square root (built on the device at init by dlog_table_entry):
//   pow[i][j], i = 0..2 : zeta^(-j * 2^(8i))          (strip byte i of the discrete logarithm)
//   pow[3][j]           : zeta^(-(j >> 1))            (half of byte 0)
//   pow[4..6][j]        : zeta^(-j * 2^(8(i-3)-1))    (half of bytes 1..3)
//   hash[low 16 bits of the canonical limb 0 of g8^j] = j,  g8 = zeta^(2^24) of order 256
struct dlog_tables {
    const uint32_t* pow;   // 7 x 256 x 9 limbs
    const uint8_t* hash;   // 65536 bytes
};
constexpr int DLOG_POW_WORDS = 7 * 256 * 9;

// the representative below q of a product output (value < 2q)
JJS_HD fe_c fq_canonical_rep(const fe_n& a) {
    fe_c t;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t d = (int32_t)a.l[i] - (int32_t)q29(i) + borrow;
        t.l[i] = (uint32_t)d & MASK29;
        borrow = d >> 29;
        if (i == 8) t.l[8] = (uint32_t)d;
    }
    fe_c r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (borrow < 0) ? a.l[i] : t.l[i];
    return r;
}
JJS_HD uint32_t dlog_key(const fe_n& x) {
    return ((fq_canonical_rep(x).l[0] * JJS_DLOG_HASH_MULT) >> JJS_DLOG_HASH_SHIFT) & 0xffffu;
}
// entry (i, j) of the power tables; also fills the hash slot of g8^j when i == 0
JJS_HD void dlog_table_entry(uint32_t* pow, uint8_t* hash, int i, int j) {
    const fe_n base = fq_as<1, 2>(fe_from_const<1, 1>(JJS_DLOG_BASES[i]));
    const int e = (i == 3) ? (j >> 1) : j;
    fe_n acc = fe_n_one();
    for (int bit = 7; bit >= 0; --bit) {
        acc = fq_sqr(acc);
        if ((e >> bit) & 1) acc = fq_mul(acc, base);
    }
    for (int k = 0; k < 9; ++k) pow[((size_t)i * 256 + j) * 9 + k] = acc.l[k];
    if (i == 0) {
        fe_n g8 = fq_as<1, 2>(fe_from_const<1, 1>(JJS_ROOT_OF_UNITY));
        for (int s2 = 0; s2 < 24; ++s2) g8 = fq_sqr(g8);
        fe_n x = fe_n_one();
        for (int bit = 7; bit >= 0; --bit) {
            x = fq_sqr(x);
            if ((j >> bit) & 1) x = fq_mul(x, g8);
        }
        hash[dlog_key(x)] = (uint8_t)j;
    }
}
JJS_HD fe_n dlog_pow(const dlog_tables& T, int i, uint32_t j) {
    fe_n r;
    const uint32_t* p = T.pow + ((size_t)i * 256 + j) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.l[k] = p[k];
    return r;
}

// r with r^2 * y == 1 when y is a non-zero square (r = 0 for y = 0).  When y is not a square the result
// is meaningless; callers verify.  w = y^((t-1)/2) gives y*w^2 = y^t =: b in the 2^32-torsion; with
// b = zeta^k the answer is w * zeta^(-k/2).  k is found a byte at a time (Sarkar / Pornin style):
// b^(2^24) lies in the order-256 subgroup, whose discrete logarithm is one table lookup.
JJS_HD fe_n fq_inv_sqrt(const fe_n& y, const dlog_tables& T) {
    const fe_n w = fq_pow_schedule(y, JJS_SQRT_SW, JJS_SQRT_SW_STEPS, JJS_SQRT_SW_TRAILING);   // y^((t-1)/2)
    const fe_n b = fq_mul(fq_mul(y, w), w);
    fe_n x = b;
    for (int i = 0; i < 24; ++i) x = fq_sqr(x);
    const uint32_t k0 = T.hash[dlog_key(x)];
    const fe_n b1 = fq_mul(b, dlog_pow(T, 0, k0));
    x = b1;
    for (int i = 0; i < 16; ++i) x = fq_sqr(x);
    const uint32_t k1 = T.hash[dlog_key(x)];
    const fe_n b2 = fq_mul(b1, dlog_pow(T, 1, k1));
    x = b2;
    for (int i = 0; i < 8; ++i) x = fq_sqr(x);
    const uint32_t k2 = T.hash[dlog_key(x)];
    const fe_n b3 = fq_mul(b2, dlog_pow(T, 2, k2));
    const uint32_t k3 = T.hash[dlog_key(b3)];
    const fe_n delta = fq_mul(fq_mul(dlog_pow(T, 3, k0), dlog_pow(T, 4, k1)), fq_mul(dlog_pow(T, 5, k2), dlog_pow(T, 6, k3)));
    return fq_mul(w, delta);
}

struct decoded_point {
    words8 u, v;     // canonical affine coordinates (identity when !ok)
    bool ok;
};

JJS_HD decoded_point decompress_point(const words8& bytes, const dlog_tables& T) {
    decoded_point out;
    words8 vw = bytes;
    const uint32_t sign = vw.w[7] >> 31;
    vw.w[7] &= 0x7fffffffu;
    bool ok = words_lt(vw, JJS_Q_WORDS);
    fe_n v = fq_from_words(vw);
    fe_n v2 = fq_sqr(v);
    auto n = fq_norm(fq_sub(v2, fq_one()));                                                   // v^2 - 1   <1,4>
    auto d = fq_norm(fq_add(fq_mul(v2, fe_from_const<1, 1>(JJS_D)), fq_one()));               // 1 + d v^2 <1,3>
    fe_n y = fq_mul(n, d);
    fe_n r = fq_inv_sqrt(y, T);
    fe_n u = fq_mul(n, r);
    // u^2 * D == N  <=>  the root exists
    ok = ok && fq_eq(fq_mul(fq_sqr(u), d), n);
    words8 uw = fq_to_words(u);
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) nz |= uw.w[i];
    const bool flip = ((uw.w[0] & 1u) ^ sign) != 0;
    ok = ok && !(nz == 0 && flip);                 // u = 0 with the sign bit set: rejected (ZIP 216)
    // -u = q - u for u != 0
    words8 neg;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)JJS_Q_WORDS[i] - uw.w[i] - borrow;
        neg.w[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t uu = (flip && nz) ? neg.w[i] : uw.w[i];
        out.u.w[i] = ok ? uu : 0u;
        out.v.w[i] = ok ? vw.w[i] : (i == 0 ? 1u : 0u);
    }
    out.ok = ok;
    return out;
}

// affine canonical (u, v) -> 32-byte compressed form
JJS_HD words8 compress_point(const words8& u, const words8& v) {
    words8 r = v;
    r.w[7] |= (u.w[0] & 1u) << 31;
    return r;
}

}  // namespace jjs
