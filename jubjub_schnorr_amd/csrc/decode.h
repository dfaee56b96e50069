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
// exponentiation y^((t-1)/2) plus the 2-adic correction, written with wave-uniform loops.
#pragma once
#include "ed29.h"

namespace jjs {

// r with r^2 * y == 1 when y is a non-zero square (r = 0 for y = 0).  When y is not a square the
// result is meaningless; callers verify.
JJS_HD fe_n fq_inv_sqrt(const fe_n& y) {
    fe_n w = fq_pow_schedule(y, JJS_SQRT_SW, JJS_SQRT_SW_STEPS, JJS_SQRT_SW_TRAILING);   // y^((t-1)/2)
    fe_n b = fq_mul(fq_mul(y, w), w);                                                    // y^t, of 2-power order
    fe_n z = fq_as<1, 2>(fe_from_const<1, 1>(JJS_ROOT_OF_UNITY));
    fe_n r = w;                      // invariant: r^2 * y * (z-power corrections) ...: r^2 * y == b^-1 * (b b^-1)
    uint32_t v = 32;
    // Invariant at the top of each pass: r^2 * y = b_0 / b ... maintained as in Tonelli-Shanks with the
    // correction applied to r = w * Z where Z^2 = 1/b at the end (b -> 1).
    for (uint32_t max_v = 32; max_v >= 1; --max_v) {
        uint32_t k = 1;
        fe_n b2k = fq_sqr(b);
        bool j_less_than_v = true;
        for (uint32_t j = 2; j < max_v; ++j) {
            const bool one = fq_is_one_weak(b2k);
            const fe_n squared = fq_sqr(fq_select(one, z, b2k));
            b2k = fq_select(one, b2k, squared);
            const fe_n new_z = fq_select(one, squared, z);
            j_less_than_v = j_less_than_v && (j != v);
            k = one ? k : j;
            z = fq_select(j_less_than_v, new_z, z);
        }
        // z now has z^2 = (generator of the subgroup b lives in)^-1-ish: multiply unless b == 1
        const bool b_one = fq_is_one_weak(b);
        const fe_n rz = fq_mul(r, z);
        r = fq_select(b_one, r, rz);
        z = fq_sqr(z);
        b = fq_mul(b, z);
        v = k;
    }
    return r;
}

struct decoded_point {
    words8 u, v;     // canonical affine coordinates (identity when !ok)
    bool ok;
};

JJS_HD decoded_point decompress_point(const words8& bytes) {
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
    fe_n r = fq_inv_sqrt(y);
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
