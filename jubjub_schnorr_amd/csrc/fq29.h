// Fq = BLS12-381 scalar field ("BlsScalar") for one-signature-per-lane kernels on gfx950.
//
// Replaces the arithmetic the reference gets from dusk-bls12_381 0.14 (Cargo.toml:24): every
// `BlsScalar` product / sum behind /root/reference/src/keys/public.rs:114-135 and behind the
// Poseidon calls at src/signatures.rs:130.
//
// Representation: 9 limbs of 29 bits in 9 VGPRs, Montgomery form with R' = 2^261.  Why not 8 x 32:
// measured on MI355X (profiles/microbench_r01.jsonl) v_mad_u64_u32 issues in ~5 cycles per wave,
// barely more than a carry add (~4.5), so instruction COUNT is what matters; with 29-bit limbs a
// whole product column (9 a*b terms + 9 m*q terms) accumulates in ONE 64-bit register through
// chained v_mad_u64_u32 with no carry instructions at all: 153 mads + ~85 shifts/masks per
// product, against ~600 instructions for the saturated 8 x 32 CIOS form.
//
// Values are kept "weakly reduced".  The C++ type carries compile-time bounds:
//     fe<L, A>:  every limb < L * 2^29,  value < A * q
// fq_mul static_asserts the two conditions that make it exact:
//     9*La*Lb*2^58 + 9*2^58 + carry < 2^64   <=  La*Lb <= 6      (column sums fit 64 bits)
//     a*b <= 70 q^2 < q * 2^261              <=  Aa*Ab <= 70     (result < 2q)
// so additions/subtractions are plain limb-wise operations with no carry propagation, and a
// result only needs `fq_norm` (carry propagation) or `fq_reduce` (conditional subtraction) when
// the type system says so.  The same source compiles for the host (tests/hostbuild) so the
// arithmetic is checked against the oracle on the CPU as well as on the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JJS_HD __host__ __device__ __forceinline__
#define JJS_CALL __host__ __device__ __attribute__((noinline))
#if defined(__HIP_DEVICE_COMPILE__)
#define JJS_CONST __constant__ const
#else
#define JJS_CONST static const
#endif
#else
#define JJS_HD inline __attribute__((always_inline))
#define JJS_CALL __attribute__((noinline))
#define JJS_CONST static const
#endif

namespace jjs {

#include "jjs_constants.inc"
#include "mont_asm.inc"

constexpr uint32_t MASK29 = 0x1fffffffu;

template <int L, int A>
struct fe {
    static_assert(L >= 1 && L <= 7, "limb bound must keep limbs below 2^32");
    static_assert(A >= 1 && A <= 70, "value bound must stay below 2^261");
    uint32_t l[9];
};
using fe_n = fe<1, 2>;   // normalised limbs, value < 2q: what every product returns
using fe_c = fe<1, 1>;   // canonical constants

struct raw9 {
    uint32_t l[9];
};

JJS_HD constexpr uint32_t q29(int i) {
    constexpr uint32_t Q[9] = {JJS_Q29_0, JJS_Q29_1, JJS_Q29_2, JJS_Q29_3, JJS_Q29_4,
                               JJS_Q29_5, JJS_Q29_6, JJS_Q29_7, JJS_Q29_8};
    return Q[i];
}
// limb i of k*q in canonical radix-2^29 form
JJS_HD constexpr uint32_t kq29(int k, int i) {
    uint64_t carry = 0;
    uint32_t out = 0;
    for (int j = 0; j <= i; ++j) {
        uint64_t t = (uint64_t)q29(j) * (uint64_t)k + carry;
        out = (uint32_t)(t & MASK29);
        carry = t >> 29;
        if (j == 8) out = (uint32_t)t;  // top limb keeps everything
    }
    return out;
}
// limb i of the subtraction pad for k*q: same value as k*q, every limb >= 2^29 - 1 (top: -1)
JJS_HD constexpr uint32_t subpad29(int k, int i) {
    return kq29(k, i) + (i < 8 ? (1u << 29) : 0u) - (i > 0 ? 1u : 0u);
}

template <int L, int A>
JJS_HD fe<L, A> fe_from_const(const uint32_t* c) {  // uniform address: scalar loads on the GPU
    fe<L, A> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c[i];
    return r;
}
JJS_HD fe_c fq_zero() {
    fe_c r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}
JJS_HD fe_c fq_one() { return fe_from_const<1, 1>(JJS_ONE); }

// widen the static bounds (no code)
template <int L2, int A2, int L, int A>
JJS_HD fe<L2, A2> fq_as(const fe<L, A>& a) {
    static_assert(L2 >= L && A2 >= A, "bounds can only be widened");
    fe<L2, A2> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i];
    return r;
}

template <int La, int Aa, int Lb, int Ab>
JJS_HD fe<La + Lb, Aa + Ab> fq_add(const fe<La, Aa>& a, const fe<Lb, Ab>& b) {
    fe<La + Lb, Aa + Ab> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <int L, int A>
JJS_HD fe<2 * L, 2 * A> fq_dbl(const fe<L, A>& a) {
    fe<2 * L, 2 * A> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] << 1;
    return r;
}
// a - b (mod q) as a + (pad - b); b must have normalised limbs.  pad = (Ab+1) q >= b limb-wise.
template <int La, int Aa, int Ab>
JJS_HD fe<La + 2, Aa + Ab + 1> fq_sub(const fe<La, Aa>& a, const fe<1, Ab>& b) {
    fe<La + 2, Aa + Ab + 1> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + (subpad29(Ab + 1, i) - b.l[i]);
    return r;
}
template <int Ab>
JJS_HD fe<2, Ab + 1> fq_neg(const fe<1, Ab>& b) {
    fe<2, Ab + 1> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = subpad29(Ab + 1, i) - b.l[i];
    return r;
}
// carry propagation: limbs back below 2^29, value unchanged
template <int L, int A>
JJS_HD fe<1, A> fq_norm(const fe<L, A>& a) {
    fe<1, A> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t t = a.l[i] + carry;
        r.l[i] = t & MASK29;
        carry = t >> 29;
    }
    r.l[8] = a.l[8] + carry;
    return r;
}
// value < 4q  ->  value < 2q  (one conditional subtraction of 2q), limbs normalised in and out
template <int A>
JJS_HD fe_n fq_reduce(const fe<1, A>& a) {
    static_assert(A <= 4, "fq_reduce handles values below 4q");
    fe_n t;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t d = (int32_t)a.l[i] - (int32_t)kq29(2, i) + borrow;
        t.l[i] = (uint32_t)d & MASK29;
        borrow = d >> 29;
        if (i == 8) t.l[8] = (uint32_t)d;
    }
    bool neg = borrow < 0;
    fe_n r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg ? a.l[i] : t.l[i];
    return r;
}
template <int L, int A>
JJS_HD fe<L, A> fq_select(bool c, const fe<L, A>& a, const fe<L, A>& b) {  // c ? a : b
    fe<L, A> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

// ---------------------------------------------------------------------------------------------
// Montgomery product (a*b - M*q) / 2^261, column by column, in a SIGNED 64-bit accumulator.
// q = 1 (mod 2^29), so the digit that clears column k is simply m_k = acc mod 2^29 when m_k*q is
// SUBTRACTED: acc - m_k has its low 29 bits clear and the carry is the arithmetic shift acc >> 29 -- one mask
// and one shift per column (adding m_k*q instead needs m_k = -acc mod 2^29: a negation, and a rounding add).
// Subtracting in all nine columns could leave the result negative (down to -q), so the LAST digit goes the
// other way: m_8 = 2^29 - (acc mod 2^29), in [1, 2^29], and m_8*q is added.  Then
//     result = (a*b - sum_{k<8} m_k q 2^(29k) + m_8 q 2^232) / 2^261   lies in (0, a*b/2^261 + q],
// the same range as the all-additive form, so the value bounds of fe<L, A> are unchanged.  The signed
// accumulator holds at most 9 products of two limbs on the positive side and 8 digit products (< 2^58 each)
// on the negative side: La*Lb*9*2^58 + carry < 2^63 needs La*Lb <= 3 (squares: L = 1).
// ---------------------------------------------------------------------------------------------
JJS_HD int64_t mont_ashr29(int64_t v) {   // floor(v / 2^29), also for negative v
    return v >> 29;
}

JJS_HD raw9 mont_mul_body(const uint32_t* a, const uint32_t* b) {
    int64_t m[9];
    raw9 r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (int64_t)((uint64_t)a[i] * b[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) acc -= m[i] * (int64_t)q29(k - i);
        if (k < 8) {
            m[k] = acc & (int64_t)MASK29;
            acc = mont_ashr29(acc);                       // (acc - m_k) / 2^29
        } else {
            m[8] = (acc & (int64_t)MASK29) - ((int64_t)1 << 29);   // minus the digit: -m_8 in [-2^29, -1]
            acc = mont_ashr29(acc - m[8]);                // (acc + m_8) / 2^29
        }
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (int64_t)((uint64_t)a[i] * b[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc -= m[i] * (int64_t)q29(k - i);   // i = 8: minus a negative digit
        r.l[k - 9] = (uint32_t)(acc & (int64_t)MASK29);
        acc = mont_ashr29(acc);
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
// square: the 36 off-diagonal products are taken once against a doubled operand
JJS_HD raw9 mont_sqr_body(const uint32_t* a) {
    int64_t m[9];
    uint32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a[i] << 1;
    raw9 r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; ++i) acc += (int64_t)((uint64_t)a[i] * a2[k - i]);
        if ((k & 1) == 0) acc += (int64_t)((uint64_t)a[k / 2] * a[k / 2]);
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc -= m[i] * (int64_t)q29(k - i);
            if (k < 8) {
                m[k] = acc & (int64_t)MASK29;
                acc = mont_ashr29(acc);
            } else {
                m[8] = (acc & (int64_t)MASK29) - ((int64_t)1 << 29);
                acc = mont_ashr29(acc - m[8]);
            }
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc -= m[i] * (int64_t)q29(k - i);
            r.l[k - 9] = (uint32_t)(acc & (int64_t)MASK29);
            acc = mont_ashr29(acc);
        }
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// Out-of-line entry points: 18 scalar arguments travel in v0..v17, the result in v0..v8.  Keeping
// the ~190-instruction body out of line bounds the kernel's code size (instruction cache) at the
// price of ~25 register moves per call.
JJS_CALL raw9 mont_mul_call(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                            uint32_t a6, uint32_t a7, uint32_t a8, uint32_t b0, uint32_t b1, uint32_t b2,
                            uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JJS_NO_MONT_ASM)
    // one hand-scheduled block (tools/gen_mont_asm.py): 153 multiply-adds in a single accumulator chain +
    // 36 masks/shifts; the result limb j overwrites a_j (hence early-clobber: no b_k may share a register with an a_j)
    asm(JJS_MONT_MUL_ASM
        : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3), [a4] "+&v"(a4), [a5] "+&v"(a5), [a6] "+&v"(a6),
          [a7] "+&v"(a7), [a8] "+&v"(a8)
        : [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [b4] "v"(b4), [b5] "v"(b5), [b6] "v"(b6), [b7] "v"(b7),
          [b8] "v"(b8)
        : JJS_MONT_ASM_CLOBBERS);
    return raw9{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
#else
    const uint32_t a[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8};
    const uint32_t b[9] = {b0, b1, b2, b3, b4, b5, b6, b7, b8};
    return mont_mul_body(a, b);
#endif
}
JJS_CALL raw9 mont_sqr_call(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                            uint32_t a6, uint32_t a7, uint32_t a8) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JJS_NO_MONT_ASM)
    asm(JJS_MONT_SQR_ASM
        : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3), [a4] "+&v"(a4), [a5] "+&v"(a5), [a6] "+&v"(a6),
          [a7] "+&v"(a7), [a8] "+&v"(a8)
        :
        : JJS_MONT_ASM_CLOBBERS);
    return raw9{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
#else
    const uint32_t a[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8};
    return mont_sqr_body(a);
#endif
}

// The same blocks inlined at the call site (no argument shuffling, no jump): for the hottest loops only --
// every inlined copy is ~1.7 KB of code.  Measured on MI355X: inlining the products of the doubling and
// addition formulas is worth 2.8 % of the verify kernel; define JJS_NO_INLINE_HOT_MUL to go back to calls.
#if !defined(JJS_NO_INLINE_HOT_MUL)
#define JJS_INLINE_HOT_MUL 1
#endif
JJS_HD raw9 mont_mul_inl(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6,
                         uint32_t a7, uint32_t a8, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4,
                         uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JJS_NO_MONT_ASM)
    asm(JJS_MONT_MUL_ASM
        : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3), [a4] "+&v"(a4), [a5] "+&v"(a5), [a6] "+&v"(a6),
          [a7] "+&v"(a7), [a8] "+&v"(a8)
        : [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [b4] "v"(b4), [b5] "v"(b5), [b6] "v"(b6), [b7] "v"(b7),
          [b8] "v"(b8)
        : JJS_MONT_ASM_CLOBBERS);
    return raw9{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
#else
    return mont_mul_call(a0, a1, a2, a3, a4, a5, a6, a7, a8, b0, b1, b2, b3, b4, b5, b6, b7, b8);
#endif
}
JJS_HD raw9 mont_sqr_inl(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6,
                         uint32_t a7, uint32_t a8) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JJS_NO_MONT_ASM)
    asm(JJS_MONT_SQR_ASM
        : [a0] "+&v"(a0), [a1] "+&v"(a1), [a2] "+&v"(a2), [a3] "+&v"(a3), [a4] "+&v"(a4), [a5] "+&v"(a5), [a6] "+&v"(a6),
          [a7] "+&v"(a7), [a8] "+&v"(a8)
        :
        : JJS_MONT_ASM_CLOBBERS);
    return raw9{{a0, a1, a2, a3, a4, a5, a6, a7, a8}};
#else
    return mont_sqr_call(a0, a1, a2, a3, a4, a5, a6, a7, a8);
#endif
}
template <int La, int Aa, int Lb, int Ab>
JJS_HD fe_n fq_mul_hot(const fe<La, Aa>& a, const fe<Lb, Ab>& b) {
    static_assert(La * Lb <= 3 && Aa * Ab <= 70, "see fq_mul");
#if defined(JJS_INLINE_HOT_MUL)
    raw9 r = mont_mul_inl(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8],
                          b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
#else
    raw9 r = mont_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8],
                           b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
#endif
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}
template <int La, int Aa>
JJS_HD fe_n fq_sqr_hot(const fe<La, Aa>& a) {
    static_assert(La == 1 && Aa * Aa <= 70, "see fq_sqr");
#if defined(JJS_INLINE_HOT_MUL)
    raw9 r = mont_sqr_inl(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8]);
#else
    raw9 r = mont_sqr_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8]);
#endif
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}

// Square of (normalised limbs + a Hades round constant) without the carry pass that the general rule La == 1
// would demand: the constants are public, and tools/gen_constants.py checks at generation time that for every
// one of them the 9-term column sums of this square stay below 28 * 2^58 (the accumulator holds 32 * 2^58).
// Only hades29.h may call it.
template <int Aa>
JJS_HD fe_n fq_sqr_plus_const(const fe<2, Aa>& a) {
    static_assert(Aa * Aa <= 70, "see fq_sqr");
    raw9 r = mont_sqr_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8]);
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}

// The products of the cooperative hash (hades29.h, coop >= 0; nobody else may call them) with the blocks inlined: that path
// is a single dependent chain on which a whole small call waits, and a call costs ~25 register moves each way.  The operands
// may be (normalised limbs + a Hades round constant): a square of such a value has the column sums of fq_sqr_plus_const
// (checked for every constant at generation time), its product with itself the same sums, its product with a value of
// normalised limbs sums below 9 * 2^59.
template <int La, int Aa, int Lb, int Ab>
JJS_HD fe_n fq_mul_chain(const fe<La, Aa>& a, const fe<Lb, Ab>& b) {
    static_assert(La * Lb <= 4 && Aa * Ab <= 70, "see fq_mul / fq_mul_plus_const");
    raw9 r = mont_mul_inl(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8],
                          b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}
template <int La, int Aa>
JJS_HD fe_n fq_sqr_chain(const fe<La, Aa>& a) {
    static_assert(La <= 2 && Aa * Aa <= 70, "see fq_sqr / fq_sqr_plus_const");
    raw9 r = mont_sqr_inl(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8]);
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}

template <int La, int Aa, int Lb, int Ab>
JJS_HD fe_n fq_mul(const fe<La, Aa>& a, const fe<Lb, Ab>& b) {
    static_assert(La * Lb <= 3, "product columns would overflow the signed 64-bit accumulator: normalise an operand");
    static_assert(Aa * Ab <= 70, "product would not reduce below 2q: reduce an operand");
    raw9 r = mont_mul_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8],
                           b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8]);
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}
template <int La, int Aa>
JJS_HD fe_n fq_sqr(const fe<La, Aa>& a) {
    static_assert(La == 1, "square columns would overflow the signed 64-bit accumulator: normalise the operand");
    static_assert(Aa * Aa <= 70, "square would not reduce below 2q");
    raw9 r = mont_sqr_call(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8]);
    fe_n o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = r.l[i];
    return o;
}

// Sum of K products with ONE Montgomery reduction: (sum_j c_j * s_j) / 2^261.  c_j are canonical
// constants read through wave-uniform addresses (SGPR operands of the mads).  K*9 + 9 <= 64 columns
// terms fit a 64-bit accumulator for K <= 6; value bound K * 1 * As <= 70.
template <int K, int As>
JJS_HD fe_n fq_dot_const(const uint32_t (*c)[9], const fe<1, As>* s) {
    static_assert(K * 9 + 9 < 64 && K * As <= 70, "dot product too long");
    uint32_t m[9];
    fe_n r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (uint64_t)c[j][i] * s[j].l[k - i];
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * q29(k - i);
            m[k] = (0u - (uint32_t)acc) & MASK29;
            acc += m[k];
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * q29(k - i);
            r.l[k - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// (sum_j s[j] * t[j]) / 2^29 mod q for small public scalars s[j] < 2^17 (wave-uniform, SGPR operands):
// K*9 small-scalar multiply-adds plus ONE Montgomery row (the quotient digit of the lowest column).
// Columns stay below K*L*2^46 + 2^58; the result is below (K * 2^17 * A / 2^29 + 1) q < 2q.
template <int K, int L, int A>
JJS_HD fe_n fq_lincomb_small(const uint32_t* s, const fe<L, A>* t) {
    static_assert(K * L <= 1024 && K * A <= 2048, "small linear combination out of range");
    fe_n r;
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) acc += (uint64_t)s[j] * t[j].l[0];
    const uint32_t m = (0u - (uint32_t)acc) & MASK29;
    acc += m;
    acc >>= 29;
#pragma unroll
    for (int k = 1; k < 9; ++k) {
#pragma unroll
        for (int j = 0; j < K; ++j) acc += (uint64_t)s[j] * t[j].l[k];
        acc += (uint64_t)m * q29(k);
        r.l[k - 1] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// ---------------------------------------------------------------------------------------------
// conversions between 8 x 32-bit canonical words and the internal form
// ---------------------------------------------------------------------------------------------
struct words8 {
    uint32_t w[8];
};

JJS_HD bool words_lt(const words8& a, const uint32_t* mod) {  // a < mod
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.w[i] - mod[i] - borrow;
        borrow = (uint32_t)(d >> 63);
    }
    return borrow != 0;
}
JJS_HD fe_c words_to_limbs(const words8& a) {
    fe_c r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t v = a.w[w];
        if (w + 1 < 8) v |= (uint64_t)a.w[w + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & MASK29;
    }
    return r;
}
// canonical words (caller has checked < q) -> Montgomery form
JJS_HD fe_n fq_from_words(const words8& a) { return fq_mul(words_to_limbs(a), fe_from_const<1, 1>(JJS_R2)); }

// Montgomery form -> fully reduced integer in [0, q), as limbs
template <int L, int A>
JJS_HD fe_c fq_canon_limbs(const fe<L, A>& a) {
    fe_c one = fq_zero();
    one.l[0] = 1;
    fe_n y = fq_mul(fq_norm(a), one);  // (a - M q) / R' with M in (-R', R'): in [1, q] since a < R'
    // y in [1, q]: q stands for zero
    bool is_q = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) is_q = is_q && (y.l[i] == q29(i));
    fe_c r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = is_q ? 0u : y.l[i];
    return r;
}
template <int L, int A>
JJS_HD words8 fq_to_words(const fe<L, A>& a) {
    fe_c c = fq_canon_limbs(a);
    words8 o;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        int bit = 32 * w, i = bit / 29, sh = bit - 29 * i;  // word w starts inside limb i
        uint64_t v = (uint64_t)c.l[i] >> sh;
        int have = 29 - sh;
        v |= (uint64_t)c.l[i + 1] << have;
        if (have + 29 < 32 && i + 2 < 9) v |= (uint64_t)c.l[i + 2] << (have + 29);
        o.w[w] = (uint32_t)v;
    }
    return o;
}
template <int L, int A>
JJS_HD bool fq_is_zero(const fe<L, A>& a) {
    fe_c c = fq_canon_limbs(a);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) x |= c.l[i];
    return x == 0;
}
template <int La, int Aa, int Ab>
JJS_HD bool fq_eq(const fe<La, Aa>& a, const fe<1, Ab>& b) {
    return fq_is_zero(fq_sub(a, b));
}

// a^e for a public exponent given as a sliding-window schedule (tools/gen_constants.py): per step
// `sq` squarings then a product with a^d, d in {1, 3, 5, 7}; finally `trailing` squarings.  Control
// flow is wave-uniform (the exponent is public).
JJS_HD fe_n fq_pow_schedule(const fe_n& a, const uint32_t (*sched)[2], int steps, int trailing) {
    fe_n a2 = fq_sqr(a);
    fe_n a3 = fq_mul(a2, a), a5 = fq_mul(a3, a2), a7 = fq_mul(a5, a2);
    fe_n e = a;
    for (int st = 0; st < steps; ++st) {
        const uint32_t nsq = sched[st][0], dg = sched[st][1];
        if (st != 0)
            for (uint32_t j = 0; j < nsq; ++j) e = fq_sqr(e);
        const fe_n m = fq_select(dg == 1, a, fq_select(dg == 3, a3, fq_select(dg == 5, a5, a7)));
        e = (st == 0) ? m : fq_mul(e, m);
    }
    for (int j = 0; j < trailing; ++j) e = fq_sqr(e);
    return e;
}
}  // namespace jjs
