// 1 / a in Fq without a 300-product power: the Bernstein-Yang "safegcd" division steps ("Fast constant-time gcd computation and
// modular inversion", 2019), in the 32-bit arrangement that is commonly used for it: nine signed limbs of 30 bits, 20 batches
// of 30 division steps (600 >= the 590 the paper's bound gives for inputs below 2^256 with the half-step start), each batch
// decided on the low 30 bits of f and g alone and applied to the full numbers as one 2 x 2 integer matrix.  Branch-free and
// the same for every lane (what a wave needs), ~11 k integer instructions against ~54 k for the addition chain of a^(q-2).
//
// Replaces the inversions behind the reference's `to_hash_inputs()` / `JubJubAffine::from` (src/signatures.rs:127-128): the
// normalisation of extended-coordinate inputs (normalize.h), the affine outputs of signing and of the multisignature sums.
// Input and output are Montgomery forms (fq29.h); 0 maps to 0, as with the power.
#pragma once
#include "fq29.h"

namespace jjs {

// bit `j` of q, from the generated 29-bit limbs
JJS_HD constexpr uint32_t q_bit(int j) { return j < 261 ? (q29(j / 29) >> (j % 29)) & 1u : 0u; }
// limb i of q in radix 2^30
JJS_HD constexpr int32_t q30(int i) {
    uint32_t v = 0;
    for (int b = 0; b < 30; ++b) v |= q_bit(30 * i + b) << b;
    return (int32_t)v;
}
// q^-1 mod 2^30 (Newton steps on the low limb; q is odd)
JJS_HD constexpr uint32_t q_inv30() {
    const uint32_t q0 = (uint32_t)q30(0) | ((uint32_t)q30(1) << 30);
    uint32_t x = q0;                                  // correct to 3 bits
    for (int i = 0; i < 5; ++i) x *= 2u - q0 * x;     // 6, 12, 24, 48 bits
    return x & 0x3fffffffu;
}
static_assert(((uint32_t)q30(0) * q_inv30() & 0x3fffffffu) == 1u, "q * q_inv30 == 1 (mod 2^30)");
static_assert(q30(8) > 0 && q30(8) < (1 << 15), "q has 255 bits: the top limb of nine holds bits 240..254");

struct s30 {
    int32_t v[9];        // value = sum v[i] 2^(30 i); limbs 0..7 in [0, 2^30) after an update, the top one signed
};
struct divstep_matrix {
    int32_t u, v, q, r;  // 2^30 (f', g') = (u f + v g, q f + r g)
};

JJS_HD int64_t s30_ashr30(int64_t x) {   // floor(x / 2^30), also for negative x
#if defined(__HIP_DEVICE_COMPILE__)
    return x >> 30;
#else
    return x >= 0 ? x >> 30 : -((-x + ((int64_t(1) << 30) - 1)) >> 30);
#endif
}

// 30 division steps on the low bits of f (odd) and g; zeta = -(delta + 1/2) of the paper's half-step variant
JJS_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, divstep_matrix& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t c1 = (uint32_t)(zeta >> 31);          // all ones when zeta < 0
        const uint32_t c2 = 0u - (g & 1u);             // all ones when g is odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;     // f, u, v negated when zeta < 0
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;                                       // zeta < 0 and g odd: the step that swaps
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;      // -zeta - 2, or zeta - 1
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}

// (d, e) <- t (d, e) / 2^30 mod q, with d, e kept in (-2q, q)
JJS_HD void update_de_30(s30& d, s30& e, const divstep_matrix& t) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);       // + q for every negative operand
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
    // the multiples of q that clear the low 30 bits
    md -= (int32_t)((q_inv30() * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((q_inv30() * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)q30(0) * md;
    ce += (int64_t)q30(0) * me;
    cd = s30_ashr30(cd); ce = s30_ashr30(ce);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        di = d.v[i]; ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)q30(i) * md;
        ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)q30(i) * me;
        d.v[i - 1] = (int32_t)cd & M30; cd = s30_ashr30(cd);
        e.v[i - 1] = (int32_t)ce & M30; ce = s30_ashr30(ce);
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}
// (f, g) <- t (f, g) / 2^30 (exact: the steps were chosen so)
JJS_HD void update_fg_30(s30& f, s30& g, const divstep_matrix& t) {
    constexpr int32_t M30 = 0x3fffffff;
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
    cf = s30_ashr30(cf); cg = s30_ashr30(cg);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        fi = f.v[i]; gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf = s30_ashr30(cf);
        g.v[i - 1] = (int32_t)cg & M30; cg = s30_ashr30(cg);
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}
// d in (-2q, q) -> [0, q), negated first when `sign` is negative
JJS_HD void normalize_30(s30& d, int32_t sign) {
    constexpr int32_t M30 = 0x3fffffff;
    int32_t r[9];
    const int32_t add1 = d.v[8] >> 31, negate = sign >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r[i] = d.v[i] + (q30(i) & add1);
        r[i] = (r[i] ^ negate) - negate;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
    const int32_t add2 = r[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] += q30(i) & add2;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
#pragma unroll
    for (int i = 0; i < 9; ++i) d.v[i] = r[i];
}

// the integer x in [0, q), given as nine canonical 29-bit limbs, in 30-bit limbs, and back
JJS_HD s30 s30_from_limbs29(const fe_c& a) {
    s30 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        // bits [30 i, 30 i + 30): from 29-bit limbs lo = (30 i) / 29 and the next one or two
        const int bit = 30 * i, lo = bit / 29, sh = bit % 29;
        uint64_t w = (uint64_t)a.l[lo] >> sh;
        if (lo + 1 < 9) w |= (uint64_t)a.l[lo + 1] << (29 - sh);
        if (lo + 2 < 9 && 58 - sh < 30) w |= (uint64_t)a.l[lo + 2] << (58 - sh);
        r.v[i] = (int32_t)((uint32_t)w & 0x3fffffffu);
    }
    return r;
}
JJS_HD fe_c limbs29_from_s30(const s30& a) {
    fe_c r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, lo = bit / 30, sh = bit % 30;
        uint64_t w = (uint64_t)(uint32_t)a.v[lo] >> sh;
        if (lo + 1 < 9) w |= (uint64_t)(uint32_t)a.v[lo + 1] << (30 - sh);
        r.l[i] = (uint32_t)w & MASK29;
    }
    return r;
}

// 1 / a (Montgomery form in, Montgomery form out; 0 for 0)
JJS_HD fe_n fq_inverse(const fe_n& a) {
    s30 d{}, e{}, f, g = s30_from_limbs29(fq_canon_limbs(a));      // g = the value a itself (out of Montgomery form), in [0, q)
#pragma unroll
    for (int i = 0; i < 9; ++i) f.v[i] = q30(i);
    e.v[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int batch = 0; batch < 20; ++batch) {
        divstep_matrix t;
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        update_de_30(d, e, t);
        update_fg_30(f, g, t);
    }
    // g == 0 and f == +-1 now (f == +-q for a == 0, and then d == 0); d = +-1/a
    normalize_30(d, f.v[8]);
    return fq_mul(limbs29_from_s30(d), fe_from_const<1, 1>(JJS_R2));      // back into Montgomery form
}

}  // namespace jjs
