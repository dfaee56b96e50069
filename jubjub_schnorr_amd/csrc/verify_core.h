// Per-signature verification logic shared by the three schemes, one signature per lane.
//
// Mirrors (same checks, same precedence, same results; different method):
//   PublicKey::verify          /root/reference/src/keys/public.rs:114-135, is_valid :159-164
//   Signature::is_valid        src/signatures.rs:93-98 ; challenge_hash :122-140
//   PublicKeyDouble::verify    src/keys/public/double.rs:86-117, is_valid :145-157
//   SignatureDouble::is_valid  src/signatures/double.rs:108-119 ; challenge_hash :151-177
//   PublicKeyVarGen::verify    src/keys/public/var_gen.rs:107-133, is_valid :160-172
//   challenge_hash (var-gen)   src/signatures/var_gen.rs:121-142
// The three schemes differ only in data: which points are validated, which field elements make
// up the challenge transcript, and which (generator, public key, R) triples must satisfy
// u*Gen + c*PK == R.  `verify_params` carries those lists as wave-uniform descriptors, so one
// kernel serves all three and nothing about a scheme is compiled in.
//
// Method (bit-exact in RESULT, not in method):
//   * fixed generator: half-size scalars (a = b*c mod r, |a|,|b| < 2^126 from a truncated Euclid) turn
//     the equation into (b*u)*G + a*PK - b*R == O: 124 shared doublings, signed 4-bit windows over two
//     per-lane tables {0..8}*PK, {0..8}*R in a global-memory workspace (144 B per entry), and a 16-bit
//     fixed-base comb for G / G' (16 mixed additions gathered from a 117 MB table).
//   * var-gen: u*Gen + c*PK by Straus over two per-lane tables, 252 shared doublings.
//   * subgroup check: order-8 Tate pairing residue test instead of [r]P -- one exponentiation per
//     fixed-generator equation (on a combination of its two points) in the first pass, one per point in
//     the resolve pass for the items the first pass cannot decide (see verify_item).
#pragma once
#include <cmath>

#include "ed29.h"
#include "hades29.h"
#include "fq_inv.h"

namespace jjs {

enum : uint32_t { ST_OK = 0, ST_INVALID_POINT = 1, ST_INVALID_SIGNATURE = 2, ST_MALFORMED = 3 };
// Internal results of the first pass (never written to the caller's status array): the item's points are
// on the curve and not the identity, but the first pass did not prove all of them torsion-free; the
// resolve pass tests each of them and turns the code into 0, 1 or 2 (see verify_item / resolve_item).
enum : uint32_t { ST_PENDING_EQ_FAILED = 4, ST_PENDING_EQ_HELD = 5 };

struct fe_src {            // where transcript element / coordinate e of item i lives: base + i*stride + off
    const uint8_t* base;
    uint32_t stride;
    uint32_t off;
};
struct eq_desc {           // u*Gen + c*PK == R
    const uint32_t* comb;  // fixed-base comb table for Gen, or nullptr when Gen is per-item data
    fe_src gen;            // affine generator (u at off, v at off+32); used when comb == nullptr
    fe_src pk;
    fe_src r;
    int32_t pk_col, gen_col;   // key-table path (key_tables.h): which deduplicated key column PK / Gen is (-1: none)
};
struct verify_params {
    uint32_t n_hash, n_points, n_eq;
#if defined(JJS_PROFILING)
    uint32_t skip_phases;            // profiling build only (bit0 validity, bit1 challenge, bit2 equations, bit3 Euclid)
#else
    uint32_t pad0_;                  // the product build has no ablation switch: see JJS_SKIP below
#endif
    fe_src hash_in[10];
    fe_src points[4];
    eq_desc eq[2];
    fe_src u;
    uint64_t n;
    uint8_t* status;                 // n bytes, or nullptr
    unsigned long long* tally;       // 4 counters, or nullptr
    uint8_t* c_out;                  // n x 32 bytes challenge (debug export), or nullptr
    const uint8_t* pre_malformed;    // n bytes from the wire decoder (non-zero: an encoding was rejected), or nullptr
    uint32_t* workspace;             // WS_WORDS_PER_LANE words per resident lane
    uint32_t own_test_mask;          // bit k: points[k] gets its own subgroup test in the first pass
    uint32_t resolve_lanes;          // lanes per queued item in the resolve pass: 1, 2 or 4 >= points left to test
    uint32_t resolve_lanes_keyed;    // ... on the key-table path, where the keys have been tested per key (only the R points are left)
    uint32_t decoded_points;         // non-zero: every point was produced by decompress_point (on the curve)
    uint32_t key_points_mask;        // bit k: points[k] is a key column of the key-table path (validated once per key)
    const uint32_t* key_flag;        // device word, non-zero when this batch runs the key-table path (key_tables.h), or nullptr
    uint32_t small_mode;             // non-zero: latency path (small_batch.h): prepare_item leaves every point check to the
                                     // per-point lanes of that path and runs no subgroup test itself
    uint8_t* prep;                   // 65 n bytes: what prepare_kernel hands to verify_kernel (see prep_record)
    uint64_t* pending;               // queue of items left to the resolve pass: item << 1 | equations held
    unsigned long long* pending_count;
};

// Ablation switches for tools/phase_profile.py exist only in the profiling build (libjjs_gpu_prof.so); in the
// product library the test is the constant `false`, so no value of any field can turn a check off.
#if defined(JJS_PROFILING)
#define JJS_SKIP(P, bits) (((P).skip_phases & (bits)) != 0u)
#else
#define JJS_SKIP(P, bits) false
#endif

constexpr int TABLE_ENTRIES = 9;                 // {0..8} * P
constexpr int ENTRY_WORDS = 36;                  // 4 coordinates x 9 limbs
constexpr int TABLE_WORDS = TABLE_ENTRIES * ENTRY_WORDS;
constexpr int WS_WORDS_PER_LANE = 2 * TABLE_WORDS;
// Fixed-base comb: the 256-bit scalar is cut into digits of JJS_COMB_BITS bits; one table row per digit position.
#ifndef JJS_COMB_BITS
#define JJS_COMB_BITS 16
#endif
constexpr int COMB_BITS = JJS_COMB_BITS;
static_assert(COMB_BITS == 8 || COMB_BITS == 16, "digits must tile a 32-bit word");
constexpr int COMB_WINDOWS = 256 / COMB_BITS, COMB_ENTRIES = 1 << COMB_BITS;
constexpr int COMB_ENTRY_WORDS = 28;             // 3 coordinates x 9 limbs, padded to 7 x 16 B
constexpr size_t COMB_TABLE_WORDS = (size_t)COMB_WINDOWS * COMB_ENTRIES * COMB_ENTRY_WORDS;

struct alignas(16) u32x4 {
    uint32_t x, y, z, w;
};

JJS_HD words8 load_words(const fe_src& s, uint64_t item, uint32_t extra_off = 0) {
    const u32x4* p = reinterpret_cast<const u32x4*>(s.base + item * s.stride + s.off + extra_off);
    u32x4 lo = p[0], hi = p[1];
    words8 r;
    r.w[0] = lo.x; r.w[1] = lo.y; r.w[2] = lo.z; r.w[3] = lo.w;
    r.w[4] = hi.x; r.w[5] = hi.y; r.w[6] = hi.z; r.w[7] = hi.w;
    return r;
}
JJS_HD void store_words(uint8_t* base, uint64_t item, const words8& v) {
    u32x4* p = reinterpret_cast<u32x4*>(base + item * 32);
    p[0] = u32x4{v.w[0], v.w[1], v.w[2], v.w[3]};
    p[1] = u32x4{v.w[4], v.w[5], v.w[6], v.w[7]};
}
JJS_HD fe_n load_fq(const fe_src& s, uint64_t item, uint32_t extra_off = 0) {
    return fq_from_words(load_words(s, item, extra_off));
}

// ---- cached-addend tables ------------------------------------------------------------------
JJS_HD void store_niels(uint32_t* dst, const niels_pt& n) {
    uint32_t w[ENTRY_WORDS];
#pragma unroll
    for (int i = 0; i < 9; ++i) { w[i] = n.ypx.l[i]; w[9 + i] = n.ymx.l[i]; w[18 + i] = n.z.l[i]; w[27 + i] = n.t2d.l[i]; }
    u32x4* p = reinterpret_cast<u32x4*>(dst);
#pragma unroll
    for (int i = 0; i < ENTRY_WORDS / 4; ++i) p[i] = u32x4{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
}
JJS_HD niels_pt load_niels(const uint32_t* src) {
    uint32_t w[ENTRY_WORDS];
    const u32x4* p = reinterpret_cast<const u32x4*>(src);
#pragma unroll
    for (int i = 0; i < ENTRY_WORDS / 4; ++i) { u32x4 v = p[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    niels_pt n;
#pragma unroll
    for (int i = 0; i < 9; ++i) { n.ypx.l[i] = w[i]; n.ymx.l[i] = w[9 + i]; n.z.l[i] = w[18 + i]; n.t2d.l[i] = w[27 + i]; }
    return n;
}
// table[k] = k*P for k = 0..8, P affine
JJS_HD void build_point_table(uint32_t* tab, const fe_n& u, const fe_n& v) {
    ext_pt p1 = ext_from_affine(u, v);
    niels_pt n1 = to_niels(p1);
    store_niels(tab, niels_identity());
    store_niels(tab + ENTRY_WORDS, n1);
    ext_pt acc = ext_double(p1, true);
    store_niels(tab + 2 * ENTRY_WORDS, to_niels(acc));
    for (int k = 3; k <= 8; ++k) {
        acc = ext_add_affine_niels(acc, n1.ypx, n1.ymx, n1.t2d, true);     // n1 is affine (Z = 1): 7 products
        store_niels(tab + k * ENTRY_WORDS, to_niels(acc));
    }
}

// ---- scalars ---------------------------------------------------------------------------------
// s + 0x0888...8: nibble i (i < 63) of the sum, minus 8, is signed digit i in [-8, 7]; nibble 63 is the
// (0 or 1) top digit.  Needs s < 2^252.
JJS_HD words8 recode_signed4(const words8& s) {
    words8 r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)s.w[i] + (i == 7 ? 0x08888888u : 0x88888888u) + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return r;
}
// word j of s for a wave-uniform j, as a select chain (a dynamically indexed register array
// would be demoted to scratch / LDS)
JJS_HD uint32_t word_at(const words8& s, int j) {
    uint32_t r = s.w[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) r = (j == k) ? s.w[k] : r;
    return r;
}
JJS_HD uint32_t nibble(const words8& s, int i) { return (word_at(s, i >> 3) >> ((i & 7) * 4)) & 15u; }
// acc + (digit `w` of the recoded scalar) * P; `top` is the index of the unsigned top digit; `flip` adds
// the opposite point (used for -b*R)
// one_entry: profiling ablation only (JJS_SKIP bit 4, constant false in the product): every lookup reads entry 1
JJS_HD ext_pt add_window(const ext_pt& acc, const uint32_t* tab, const words8& sc, int w, bool need_t, int top = 63,
                         bool flip = false, bool one_entry = false) {
    uint32_t nib = nibble(sc, w);
    int d = (w == top) ? (int)nib : (int)nib - 8;
    bool neg = d < 0;
    uint32_t idx = (uint32_t)(neg ? -d : d);
    idx = idx > 8u ? 8u : idx;    // only an out-of-range (malformed, status 3) scalar gets here: stay inside the table
    idx = one_entry ? 1u : idx;
    return ext_add_niels(acc, load_niels(tab + idx * ENTRY_WORDS), neg != flip, need_t);
}

// ---- arithmetic mod r (JubJubScalar), 8 x 32-bit Montgomery; only u = r - c*sk needs it ----------
JJS_HD words8 fr_mont_mul(const words8& a, const words8& b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a.w[j] * b.w[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        uint32_t mq = t[0] * JJS_FR_INV32;
        c = (uint64_t)mq * JJS_FR_WORDS[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)mq * JJS_FR_WORDS[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    words8 r, s;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.w[i] = t[i];
        uint64_t d = (uint64_t)t[i] - JJS_FR_WORDS[i] - borrow;
        s.w[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    bool keep = (t[8] == 0) && borrow;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = keep ? r.w[i] : s.w[i];
    return r;
}
// (a - b*c) mod r for canonical a, b, c
JJS_HD words8 fr_sub_mul(const words8& a, const words8& b, const words8& c) {
    words8 r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.w[i] = JJS_FR_R2_WORDS[i];
    words8 bc = fr_mont_mul(fr_mont_mul(b, r2), c);
    words8 d;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.w[i] - bc.w[i] - borrow;
        d.w[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
    uint32_t mask = 0u - borrow, carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)d.w[i] + (JJS_FR_WORDS[i] & mask) + carry;
        d.w[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    return d;
}

// ---- half-size scalars (Antipa, Brown, Gallant, Lambert, Struik, Vanstone: "Accelerated verification
// of ECDSA signatures") -------------------------------------------------------------------------------
// For the challenge c find a, b with  a = b*c (mod r),  0 <= a < 2^126,  0 < |b| < 2^126, by running
// Euclid on (r, c) until the remainder drops below 2^126.  Then, for points of order r,
//     u*G + c*PK == R   <=>   (b*u mod r)*G + a*PK - b*R == O      (b is invertible mod r),
// which needs 126 shared doublings instead of 252.  The Euclid passes use partial quotients estimated in
// double precision (half_size_scalars below); every lane runs the same loop body and lanes that have
// finished idle until the slowest lane of the wave is done.
struct u128w {
    uint32_t w[4];
};
struct half_scalars {
    u128w a, b;      // a, |b| < 2^126
    bool b_neg;
};

JJS_HD bool wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(x) != 0ull;
#else
    return x;
#endif
}
JJS_HD int bitlen256(const words8& x) {
    int bl = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bl = x.w[i] ? 32 * i + 32 - __builtin_clz(x.w[i]) : bl;
    return bl;
}
// x << d for 0 <= d < 256, barrel shifter: word moves by 4, 2, 1 words, then a sub-word funnel shift
template <int N>
JJS_HD void shl_words(uint32_t (&x)[N], int d) {
#pragma unroll
    for (int stage = 4; stage >= 1; stage >>= 1) {
        if (stage < N) {
            bool on = (d & (32 * stage)) != 0;
#pragma unroll
            for (int i = N - 1; i >= 0; --i) x[i] = on ? (i >= stage ? x[i - stage] : 0u) : x[i];
        }
    }
    int bs = d & 31;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        uint64_t v = ((uint64_t)x[i] << 32) | (i ? x[i - 1] : 0u);
        x[i] = (uint32_t)((v << bs) >> 32);
    }
}
template <int N>
JJS_HD bool lt_words(const uint32_t (&a)[N], const uint32_t (&b)[N]) {  // a < b
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        uint64_t t = (uint64_t)a[i] - b[i] - borrow;
        borrow = (uint32_t)(t >> 63);
    }
    return borrow != 0;
}

// the 64 bits of x that start at bit position `pos` (pos may be negative: zero filled)
JJS_HD uint64_t bits64_at(const uint32_t (&x)[8], int pos) {
    // assemble from three words around pos
    const int p = pos < 0 ? 0 : pos;
    const int wi = p >> 5, sh = p & 31;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w0 = (i == wi) ? x[i] : w0;
        w1 = (i == wi + 1) ? x[i] : w1;
        w2 = (i == wi + 2) ? x[i] : w2;
    }
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    uint64_t v = lo >> sh;
    if (sh) v |= (uint64_t)w2 << (64 - sh);
    return pos < 0 ? (v << (-pos)) : v;
}

// Reciprocal ESTIMATE only.  v_rcp_f64 is not correctly rounded (treat it as good to ~2^-23 relative, the
// figure older ISA manuals give); nothing below relies on its accuracy for correctness: the remainder
// xn = x - q*ys is computed exactly (fma on integers below 2^53 whose true result is below 2^53), a quotient
// that is off by one is repaired from the sign / size of xn, and a step is accepted only after the explicit
// range certificate `xn >= cn && ys - xn >= cn + c1`, which implies 0 <= xn < ys, i.e. that q IS floor(x / ys).
// A worse estimate therefore ends the lane's Lehmer run early (the caller falls back to the full-precision
// step), it cannot produce a wrong step.  tests/test_gpu_parity.py::test_half_size_scalars_on_device and the
// Euclid stage of tools/devcheck run this code path on the device against the textbook algorithm.
JJS_HD double rcp_estimate(double y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(y);
#else
    return 1.0 / y;
#endif
}

// Lehmer's inner loop: up to LEHMER_STEPS consecutive steps of Euclid's algorithm, decided on the leading
// 52 bits of the two remainders alone.  x >= y are those leading bits (r0 = x*2^p + e0, r1 = y*2^p + e1,
// 0 <= e < 2^p) held as doubles (integers below 2^53: every operation here is exact).  With cofactors
// X_i = (-1)^i (u_i r0 - v_i r1) the true remainder is X_i = x_i 2^p + d_i, |d_i| < c_i 2^p, c_i = max(u_i, v_i),
// so a step with quotient q = floor(x_{i-1} / x_i) is the true Euclid step (0 <= X_{i+1} < X_i) whenever
//     x_{i+1} >= c_{i+1}   and   x_i - x_{i+1} >= c_i + c_{i+1},
// and the remainder it divides by is still at least 2^126 (the stopping rule of half_size_scalars) whenever
//     x_i >= th + c_i,  th = 2^(126 - p)  (1 when p > 126).
// A step that cannot be certified ends the lane's run; the caller applies the certified ones to the full
// numbers and comes back with fresh leading bits (or takes one full-precision step if there were none).
constexpr int LEHMER_STEPS = 20;
struct lehmer_run {
    uint32_t u0, v0, u1, v1;   // (X_k, X_{k+1}) = +-(u0 r0 - v0 r1), -+(u1 r0 - v1 r1); all below 2^26
    uint32_t steps;            // k
};
JJS_HD lehmer_run lehmer_steps(double x, double y, double th, bool live) {
    double u0 = 1.0, v0 = 0.0, u1 = 0.0, v1 = 1.0;
    uint32_t steps = 0;
    for (int k = 0; k < LEHMER_STEPS; ++k) {
        const double c1 = u1 > v1 ? u1 : v1;
        bool ok = live && y >= th + c1;
        const double ys = ok ? y : 1.0;
        double q = ::floor(x * rcp_estimate(ys));
        double xn = ::fma(-q, ys, x);                 // exact (see rcp_estimate); one unit of error in q is repaired here
        const bool low = xn < 0.0, high = xn >= ys;
        q = low ? q - 1.0 : (high ? q + 1.0 : q);
        xn = low ? xn + ys : (high ? xn - ys : xn);
        const double un = ::fma(q, u1, u0), vn = ::fma(q, v1, v0);
        const double cn = un > vn ? un : vn;
        ok = ok && q < 67108864.0 && cn < 67108864.0 && xn >= cn && (ys - xn) >= cn + c1;
        x = ok ? ys : x;
        y = ok ? xn : y;
        u0 = ok ? u1 : u0; v0 = ok ? v1 : v0;
        u1 = ok ? un : u1; v1 = ok ? vn : v1;
        steps += ok ? 1u : 0u;
        live = ok;
    }
    lehmer_run r;
    r.u0 = (uint32_t)u0; r.v0 = (uint32_t)v0; r.u1 = (uint32_t)u1; r.v1 = (uint32_t)v1; r.steps = steps;
    return r;
}
// m1 * a - m2 * b for 8-word a, b and 32-bit m1, m2, known to lie in [0, 2^256)
JJS_HD void mul_sub_words(uint32_t (&out)[8], uint32_t m1, const uint32_t (&a)[8], uint32_t m2, const uint32_t (&b)[8]) {
    uint64_t ca = 0, cb = 0;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ca += (uint64_t)m1 * a[i];
        cb += (uint64_t)m2 * b[i];
        const uint64_t d = (uint64_t)(uint32_t)ca - (uint32_t)cb - borrow;
        out[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
        ca >>= 32; cb >>= 32;
    }
}
// m1 * a + m2 * b for 4-word a, b (the result fits 4 words)
JJS_HD void mul_add_words(uint32_t (&out)[4], uint32_t m1, const uint32_t (&a)[4], uint32_t m2, const uint32_t (&b)[4]) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t p = (uint64_t)m1 * a[i] + (uint32_t)carry;
        const uint64_t s2 = (uint64_t)m2 * b[i] + (uint32_t)p;
        out[i] = (uint32_t)s2;
        carry = (carry >> 32) + (p >> 32) + (s2 >> 32);
    }
}

// Euclid on (r, c), truncated at the first remainder below 2^126: a = that remainder, b = its cofactor of c
// (a = b*c mod r, a, |b| < 2^126).  Each pass first runs Lehmer's inner loop on the leading 52 bits (about 15
// quotients per pass at ~30 instructions each, see lehmer_steps) and applies the resulting 2x2 matrix to the
// full numbers; a lane for which no step could be certified (a quotient of 2^26 or more, a remainder within
// 2^p of 2^126: adversarial c only) takes one full-precision step instead: q' * r1 is removed from r0 for a
// partial quotient q' <= floor(r0 / r1) estimated from the leading 63 bits in double precision (clamped to 31
// significant bits, at least 1), and a swap happens only when the remainder has dropped below r1.  Every pass
// is a sequence of true Euclid steps, so the invariant r0*|t1| + r1*|t0| = r holds throughout and the result
// is the one of the textbook algorithm.  ~7 passes; the loop runs until the slowest lane of the wave is done.
JJS_HD half_scalars half_size_scalars(const words8& c) {
    uint32_t r0[8], r1[8], t0[4] = {0, 0, 0, 0}, t1[4] = {1, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) { r0[i] = JJS_FR_WORDS[i]; r1[i] = c.w[i]; }
    bool neg = false;   // sign of t1; t0 always has the opposite sign (or is zero)
    // The bound only guarantees that every wave leaves the loop whatever happens (each pass makes progress
    // in every active lane; the worst case, all quotients 1, is 185 steps).
    for (int pass = 0; pass < 512; ++pass) {
        words8 w1;
#pragma unroll
        for (int i = 0; i < 8; ++i) w1.w[i] = r1[i];
        const bool active = bitlen256(w1) > 126;
        if (!wave_any(active)) break;
        words8 w0;
#pragma unroll
        for (int i = 0; i < 8; ++i) w0.w[i] = r0[i];
        const int len0 = bitlen256(w0);
        // ---- Lehmer run on the leading 52 bits of r0 and the bits of r1 at the same position -------------
        bool single;   // this lane is active and no step could be certified: one full-precision step below
        {
            const int p = len0 - 52;                                   // >= 75: r0 >= r1 >= 2^126 in active lanes
            const double x = (double)(bits64_at(r0, len0 - 64) >> 12), y = (double)(bits64_at(r1, len0 - 64) >> 12);
            const int sh = 126 - p;
            const double th = sh > 0 ? (double)(1ull << (sh > 62 ? 62 : sh)) : 1.0;   // sh <= 51 in active lanes
            const lehmer_run L = lehmer_steps(x, y, th, active);
            const bool odd = (L.steps & 1u) != 0;
            uint32_t A[8], B[8], n0[8], n1[8], m0[4], m1[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) { A[i] = odd ? r1[i] : r0[i]; B[i] = odd ? r0[i] : r1[i]; }
            mul_sub_words(n0, odd ? L.v0 : L.u0, A, odd ? L.u0 : L.v0, B);      // X_k
            mul_sub_words(n1, odd ? L.u1 : L.v1, B, odd ? L.v1 : L.u1, A);      // X_{k+1}
            mul_add_words(m0, L.u0, t0, L.v0, t1);
            mul_add_words(m1, L.u1, t0, L.v1, t1);
#pragma unroll
            for (int i = 0; i < 8; ++i) { r0[i] = n0[i]; r1[i] = n1[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { t0[i] = m0[i]; t1[i] = m1[i]; }
            neg = odd ? !neg : neg;
            single = active && L.steps == 0;        // then the matrix was the identity: w0, w1 still hold r0, r1
        }
        if (!wave_any(single)) continue;
        // leading 63 bits of r0 and of r1, each at its own position: r0 >= A * 2^p0 and r1 < (B + 1) * 2^p1,
        // so r0 / r1 > A / (B + 1) * 2^(p0 - p1)
        const int p0 = bitlen256(w0) - 63, p1 = bitlen256(w1) - 63;
        const uint64_t A = bits64_at(r0, p0), B = bits64_at(r1, p1);
        const int e = single ? p0 - p1 : 0;                       // >= 0 because r0 >= r1
        double qd = ((double)A / ((double)B + 1.0)) * (1.0 - 1.0 / 1125899906842624.0);   // in (0.49, 2)
        // A quotient of 2^31 or more (adversarial c only) is taken as q * 2^k with q below 2^31: the
        // multiple removed is q * (r1 << k).  The shifter runs only when some lane of the wave needs it.
        uint32_t x[8], y[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = r1[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = t1[i];
        const int k = e > 30 ? e - 30 : 0;
        qd = qd * (double)(1u << (e - k));                        // * 2^min(e, 30)
        if (wave_any(k > 0)) {
            shl_words(x, k);
            shl_words(y, k);
        }
        uint32_t q = (uint32_t)qd;
        q = q ? q : 1u;                       // r0 >= r1 always, so one multiple can be removed
        q = single ? q : 0u;
        // r0 -= q * x ; t0 += q * y
        uint32_t n0[8], m0[4];
        uint64_t carry = 0;
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            carry += (uint64_t)q * x[i];
            uint64_t d = (uint64_t)r0[i] - (uint32_t)carry - borrow;
            n0[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 63);
            carry >>= 32;
        }
        carry = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            carry += (uint64_t)q * y[i] + t0[i];
            m0[i] = (uint32_t)carry;
            carry >>= 32;
        }
        const bool swap = single && lt_words(n0, r1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t a = n0[i], b = r1[i];
            r0[i] = swap ? b : a;
            r1[i] = swap ? a : b;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = m0[i], b = t1[i];
            t0[i] = swap ? b : a;
            t1[i] = swap ? a : b;
        }
        neg = swap ? !neg : neg;
    }
    half_scalars h;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h.a.w[i] = r1[i]; h.b.w[i] = t1[i]; }
    h.b_neg = neg;
    return h;
}

// [r]P == identity for affine P, double-and-add over the public bits of r (top bit 251).
// The reference's definition (dusk-jubjub `is_torsion_free`); kept as the cross-check of the
// pairing test below (tools/devcheck, debug entry point) -- the verify path uses the pairing test.
JJS_HD bool is_torsion_free_by_order(const fe_n& u, const fe_n& v) {
    ext_pt p = ext_from_affine(u, v);
    niels_pt n = to_niels(p);
    ext_pt acc = p;
    for (int i = 250; i >= 0; --i) {
        bool bit = (JJS_FR_WORDS[i >> 5] >> (i & 31)) & 1;
        acc = ext_double(acc, bit);
        if (bit) acc = ext_add_affine_niels(acc, n.ypx, n.ymx, n.t2d, false);
    }
    return ext_is_identity(acc);
}

// Same predicate for every ON-CURVE point other than the identity, at ~1/8 of the cost: P is in the
// prime-order subgroup iff the reduced order-8 Tate pairing with the generator T8 of the (cyclic)
// 2-Sylow subgroup is trivial: g(P)^((q-1)/8) == 1, with g the Miller function cleared of
// denominators modulo 8th powers (derivation and constants: tools/gen_constants.py).  g vanishes on
// the small-order points it cannot evaluate (including the identity), which yields "false"; the
// caller rejects the identity separately, as the reference does.
// Projective form: for P = (X : Y : Z) the affine expression has denominator Z^48 = (Z^6)^8, an 8th power,
// so the numerators alone give the same residue class.
JJS_HD bool pairing_is_trivial(const fe_n& px, const fe_n& py, const fe_n& pz) {
    auto opv = fq_add(pz, py);                            // Z + Y
    auto omv = fq_sub(pz, py);                            // Z - Y
    fe_n w = fq_mul(omv, px);                             // W = (Z-Y) X
    fe_n x = fq_mul(opv, px);                             // X' = (Z+Y) X
    fe_n oz = fq_mul(opv, pz);                            // (Z+Y) Z
    fe_n xw[2] = {x, w};
    auto l1 = fq_norm(fq_add(oz, fq_dot_const<2, 2>(JJS_PAIR_L1, xw)));
    auto l2 = fq_norm(fq_add(oz, fq_mul(x, fe_from_const<1, 1>(JJS_PAIR_NEG_L2))));
    auto vv = fq_norm(fq_sub(x, w));
    fe_n a = fq_mul(fq_sqr(l1), l2);
    a = fq_sqr(fq_mul(a, fq_sqr(vv)));                    // L1^4 L2^2 V^4
    fe_n y = fq_mul(fq_mul(x, w), fe_from_const<1, 1>(JJS_PAIR_B));
    fe_n y2 = fq_sqr(y);
    fe_n y7 = fq_mul(fq_mul(fq_sqr(y2), y2), y);
    fe_n g = fq_mul(a, y7);
    fe_n e = fq_pow_schedule(g, JJS_PAIR_SW, JJS_PAIR_SW_STEPS, JJS_PAIR_SW_TRAILING);
    return fq_eq(e, fq_one());
}
JJS_HD bool is_torsion_free(const fe_n& u, const fe_n& v) { return pairing_is_trivial(u, v, fe_n_one()); }

// is_on_curve && !is_identity: the cheap part of `is_valid` (src/keys/public.rs:159-164)
JJS_HD bool point_on_curve_not_identity(const fe_n& u, const fe_n& v) {
    return affine_on_curve(u, v) && !affine_is_identity(u, v);
}

// digit positions [lo, hi) of the comb; T of the result is valid only when t_last is set
JJS_HD ext_pt add_comb_range(ext_pt acc, const uint32_t* comb, const words8& k, int lo, int hi, bool t_last,
                             bool one_entry = false) {
    for (int i = lo; i < hi; ++i) {
        constexpr int per_word = 32 / COMB_BITS;
        uint32_t digit = (word_at(k, i / per_word) >> ((i % per_word) * COMB_BITS)) & (uint32_t)(COMB_ENTRIES - 1);
        digit = one_entry ? (digit & 255u) : digit;  // profiling ablation only: 256 entries per row, cache-resident
        const u32x4* p = reinterpret_cast<const u32x4*>(comb + ((size_t)i * COMB_ENTRIES + digit) * COMB_ENTRY_WORDS);
        uint32_t w[COMB_ENTRY_WORDS];
#pragma unroll
        for (int k4 = 0; k4 < COMB_ENTRY_WORDS / 4; ++k4) { u32x4 v = p[k4]; w[4 * k4] = v.x; w[4 * k4 + 1] = v.y; w[4 * k4 + 2] = v.z; w[4 * k4 + 3] = v.w; }
        fe_t ypx, ymx, t2d;
#pragma unroll
        for (int j = 0; j < 9; ++j) { ypx.l[j] = w[j]; ymx.l[j] = w[9 + j]; t2d.l[j] = w[18 + j]; }
        acc = ext_add_affine_niels(acc, ypx, ymx, t2d, t_last || i != hi - 1);
    }
    return acc;
}
JJS_HD ext_pt add_comb(ext_pt acc, const uint32_t* comb, const words8& k, bool one_entry = false) {
    return add_comb_range(acc, comb, k, 0, COMB_WINDOWS, false, one_entry);
}
JJS_HD words8 widen128(const u128w& x) {
    words8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = i < 4 ? x.w[i] : 0u;
    return r;
}
// x + 0x0888...8 over 32 nibbles: signed digits for a scalar below 2^127 (top digit 31 unsigned)
JJS_HD words8 recode_signed4_128(const u128w& s) {
    words8 r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)(i < 4 ? s.w[i] : 0u) + (i < 3 ? 0x88888888u : (i == 3 ? 0x08888888u : 0u)) + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return r;
}

JJS_HD words8 select_words(bool c, const words8& a, const words8& b) {
    words8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = c ? a.w[i] : b.w[i];
    return r;
}

// (b*u) mod r for the signed half-size scalar b: the fixed-base scalar of the half-size equation
JJS_HD words8 half_scalar_times_u(const half_scalars& h, const words8& u) {
    words8 r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.w[i] = JJS_FR_R2_WORDS[i];
    words8 w = fr_mont_mul(fr_mont_mul(widen128(h.b), r2), u);          // |b|*u mod r
    uint32_t nz = 0, borrow = 0;
    words8 n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        nz |= w.w[i];
        uint64_t d = (uint64_t)JJS_FR_WORDS[i] - w.w[i] - borrow;
        n.w[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    return select_words(h.b_neg && nz != 0, n, w);                    // b*u mod r
}

// u*Gen + c*PK == R.
//  * Fixed generator (comb table present), through half-size scalars:
//      (b*u mod r)*G + a*PK - b*R == O   with a = b*c (mod r), a, |b| < 2^126,
//    which is equivalent to the reference's equation whenever PK and R have order r -- and when they
//    do not the item is InvalidPoint whatever this returns.  Two per-lane tables (PK, R), 124 shared
//    doublings, then the comb additions for (b*u)*G.
//  * Per-item generator: Straus over two per-lane tables (PK, Gen), 252 shared doublings, compared
//    with R projectively.
// Both cases run the same window loop (one copy of the doubling and addition code in the kernel).
JJS_HD bool check_equation(const eq_desc& E, uint64_t item, uint32_t* ws, const words8& u, const words8& c,
                           const half_scalars& h, const uint32_t* shared_tab = nullptr) {
    // shared_tab: profiling ablation only (JJS_SKIP bit 4; nullptr in the product): every window lookup reads entry 1
    // of the lane's own table and every comb lookup one of 256 entries of its row, so the working set of all lanes
    // (19 MB + 0.5 MB) stays in L2 without any two lanes sharing an address (a single shared entry would measure a
    // hot spot, not a cache hit) -- what the lookups cost beyond that is the price of the gathers
    const bool one_entry = shared_tab != nullptr;
    const bool fixed = (E.comb != nullptr);
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {                       // table 0: PK; table 1: R (fixed) or Gen
        const fe_src& src = (t == 0) ? E.pk : (fixed ? E.r : E.gen);
        fe_n pu = load_fq(src, item), pv = load_fq(src, item, 32);
        build_point_table(ws + t * TABLE_WORDS, pu, pv);
    }
    words8 s0, s1, w;
    int top;
    bool flip1;
    if (fixed) {
        w = half_scalar_times_u(h, u);
        s0 = recode_signed4_128(h.a);
        s1 = recode_signed4_128(h.b);
        top = 31;
        flip1 = !h.b_neg;                                             // table 1 contributes -b*R
    } else {
        s0 = recode_signed4(c);
        s1 = recode_signed4(u);
        w = u;
        top = 63;
        flip1 = false;
    }
    ext_pt acc = ext_identity();
    for (int win = top; win >= 0; --win) {
        if (win != top) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) acc = ext_double(acc, j == 3);
        }
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            const words8 sc = select_words(t == 0, s0, s1);
            const bool need_t = (t == 0) || (fixed && win == 0);      // comb additions follow the last window
            acc = add_window(acc, ws + t * TABLE_WORDS, sc, win, need_t, top, t == 1 && flip1, one_entry);
        }
    }
    if (fixed) {
        acc = add_comb(acc, E.comb, w, one_entry);
        return ext_is_identity(acc);
    }
    fe_n ru = load_fq(E.r, item), rv = load_fq(E.r, item, 32);
    return ext_eq_affine(acc, ru, rv);
}

// One pairing test per fixed-generator equation instead of one per point.  Write PK = P + S and R = Q + T with
// P, Q of order r and S, T in the 2-Sylow subgroup (cyclic of order 8, so S and T are residues mod 8).  The
// pairing residue is a homomorphism that is trivial exactly on the prime-order subgroup, hence the test on
// W = R + k*PK says T + k*S = 0 (mod 8).  The half-size equation (b*u)G + a*PK - b*R == O says, on the torsion
// part, a*S - b*T = 0 (mod 8).  With k = 0 when a is odd and k = 1 when a is even and b odd, the determinant
// of the two relations is odd, so together they force S = T = 0: both points are torsion-free and the equation
// is the reference's.  If either test fails nothing is concluded and the item goes to the resolve pass.
// (The addition law is complete on JubJub, so W and the equation are computed exactly for any curve points.)
JJS_HD bool combined_subgroup_test(const eq_desc& E, uint64_t item, const half_scalars& h) {
    const bool a_odd = (h.a.w[0] & 1u) != 0, b_odd = (h.b.w[0] & 1u) != 0;
    fe_n ru = load_fq(E.r, item), rv = load_fq(E.r, item, 32);
    fe_n pu = load_fq(E.pk, item), pv = load_fq(E.pk, item, 32);
    niels_pt n = niels_select(!a_odd, to_niels(ext_from_affine(pu, pv)), niels_identity());
    ext_pt w = ext_add_affine_niels(ext_from_affine(ru, rv), n.ypx, n.ymx, n.t2d, false);
    return (a_odd || b_odd) && pairing_is_trivial(w.x, w.y, w.z);
}

// First pass over one signature; ws = this lane's WS_WORDS_PER_LANE workspace words.  Returns the final
// status, or ST_PENDING_* when the points' subgroup membership is still open:
//  * fixed generator (single, double): see combined_subgroup_test -- status Ok needs the equation and the
//    combined test to hold; anything else is pending;
//  * per-item generator: PK and Gen get their own tests (own_test_mask); if the equation holds, R equals
//    u*Gen + c*PK and is torsion-free with them; if it fails, R's test is pending.
// The work of a verification is cut in two so that the device can run each half at the occupancy its
// register needs allow (prepare_kernel: four waves per SIMD, verify_kernel: two; device_kernels.h):
//   prepare_item : encodings, cheap point checks, own subgroup tests, challenge, half-size scalars, combined
//                  subgroup tests -- everything except the window tables
//   finish_item  : the equations and the verdict
// prep_record is what passes between them (a device buffer of 65 bytes per item; locals on the CPU build).
struct prep_record {
    words8 c;            // challenge (250 bits)
    half_scalars h;      // a, |b|, sign of b: zero for the per-item-generator scheme
    bool malformed;      // an encoding out of range (status 3)
    bool valid;          // every point on the curve and not the identity, own subgroup tests passed
    bool proven;         // every combined subgroup test passed
};

// Does this batch run the key-table path?  Decided on the device after the keys have been counted.
JJS_HD bool keyed_mode(const verify_params& P) { return P.key_flag != nullptr && *P.key_flag != 0u; }

// The device runs a batch that may take the key-table path in two launches, because whether it does is decided while
// the first one runs (the keys are being counted beside it):
//   PREP_HEAD  everything that does not depend on the decision: encodings of the transcript, the cheap checks of the points
//              that are not keys, the challenge.  It does not read u at all (the challenge does not depend on it), so that
//              a host-buffer call can send the u column last, behind everything the hashes need (host_calls.h
//              run_host_block); the range check of u is made by whoever reads it next:
//   PREP_TAIL  what only the throughput path needs (the launch leaves at once when the key tables engaged): u < r,
//              validity of the key points, half-size scalars, combined subgroup tests; on the key-table path
//              kt_finish_item checks u < r itself;
//   PREP_ALL   both at once, with the decision already known (keyed_mode): every other caller.
enum prep_phase : int { PREP_ALL = 0, PREP_HEAD = 1, PREP_TAIL = 2 };

// validity of the points selected by `mask` (InvalidPoint takes precedence over InvalidSignature)
JJS_HD bool points_valid(const verify_params& P, uint64_t item, uint32_t mask) {
    bool valid = true;
    for (uint32_t k = 0; k < P.n_points; ++k) {
        if (!((mask >> k) & 1u)) continue;
        fe_n pu = load_fq(P.points[k], item), pv = load_fq(P.points[k], item, 32);
        // points that come out of the wire decoder satisfy the curve equation by construction
        valid = (P.decoded_points ? !affine_is_identity(pu, pv) : point_on_curve_not_identity(pu, pv)) && valid;
        if ((P.own_test_mask >> k) & 1u) valid = is_torsion_free(pu, pv) && valid;
    }
    return valid;
}
// half-size scalars (shared by both equations of the double scheme) and the combined subgroup tests
JJS_HD void scalars_and_combined_tests(const verify_params& P, uint64_t item, bool check_points, prep_record& r) {
    const uint32_t n_eq = JJS_SKIP(P, 4u) ? 0u : P.n_eq;
    half_scalars h{};
    if (n_eq && P.eq[0].comb) {
        if (JJS_SKIP(P, 8u)) {                           // profiling only: stand-in scalars, no Euclid
#pragma unroll
            for (int i = 0; i < 4; ++i) { h.a.w[i] = r.c.w[i] >> 2; h.b.w[i] = r.c.w[4 + i] >> 2; }
        } else {
            h = half_size_scalars(r.c);
        }
    }
    r.h = h;
    bool proven = true;
    for (uint32_t k = 0; k < n_eq; ++k)
        if (check_points && P.eq[k].comb) proven = combined_subgroup_test(P.eq[k], item, h) && proven;
    r.proven = proven;
}

// coop: see hades_permute (the latency path's hash lanes work in groups of eight); -1 everywhere else
JJS_HD prep_record prepare_item(const verify_params& P, uint64_t item, bool write_c = true, int coop = -1, prep_phase phase = PREP_ALL) {
    prep_record r;
    const bool keyed = phase == PREP_ALL && keyed_mode(P);
    // 1. encodings: every transcript element (all point coordinates and m) < q, u < r (not in the head launch: see prep_phase)
    const bool reads_u = phase != PREP_HEAD || JJS_SKIP(P, 2u);
    const words8 u = reads_u ? load_words(P.u, item) : words8{};
    bool malformed = phase != PREP_HEAD && !words_lt(u, JJS_FR_WORDS);
    if (P.pre_malformed) malformed = malformed || P.pre_malformed[item] != 0;
    for (uint32_t e = 0; e < P.n_hash; ++e) malformed = malformed || !words_lt(load_words(P.hash_in[e], item), JJS_Q_WORDS);
    r.malformed = malformed;

    // 2. point validity: a key is validated once per key on the key-table path, not per item
    const bool check_points = !JJS_SKIP(P, 1u) && !P.small_mode;
    const uint32_t all_points = (1u << P.n_points) - 1u;
    const uint32_t mine = (keyed || phase == PREP_HEAD) ? (all_points & ~P.key_points_mask) : all_points;
    r.valid = check_points ? points_valid(P, item, mine) : true;

    // 3. challenge
    words8 c = u;
    if (!JJS_SKIP(P, 2u)) {
        fe_n digest = poseidon_digest((int)P.n_hash, [&](int e) { return load_fq(P.hash_in[e], item); }, coop);
        c = truncate250(digest);
    }
    if (P.c_out && write_c) store_words(P.c_out, item, c);
    r.c = c;

    // 4. half-size scalars and the combined subgroup tests: the key-table path needs neither
    r.h = half_scalars{};
    r.proven = true;
    if (!keyed && phase == PREP_ALL) scalars_and_combined_tests(P, item, check_points, r);
    return r;
}
// PREP_TAIL: completes the record of PREP_HEAD for a batch that turned the key tables down
JJS_HD prep_record prepare_tail(const verify_params& P, uint64_t item, prep_record r) {
    const bool check_points = !JJS_SKIP(P, 1u) && !P.small_mode;
    r.malformed = r.malformed || !words_lt(load_words(P.u, item), JJS_FR_WORDS);
    if (check_points) r.valid = points_valid(P, item, P.key_points_mask & ((1u << P.n_points) - 1u)) && r.valid;
    scalars_and_combined_tests(P, item, check_points, r);
    return r;
}

// ws = this lane's WS_WORDS_PER_LANE workspace words
JJS_HD uint32_t finish_item(const verify_params& P, uint64_t item, uint32_t* ws, const prep_record& r) {
    const words8 u = load_words(P.u, item);
    bool eq_ok = true;
    const uint32_t n_eq = JJS_SKIP(P, 4u) ? 0u : P.n_eq;
    for (uint32_t k = 0; k < n_eq; ++k) eq_ok = check_equation(P.eq[k], item, ws, u, r.c, r.h, JJS_SKIP(P, 16u) ? P.workspace : nullptr) && eq_ok;
    if (r.malformed) return ST_MALFORMED;
    if (!r.valid) return ST_INVALID_POINT;
    if (JJS_SKIP(P, 1u)) return eq_ok ? ST_OK : ST_INVALID_SIGNATURE;
    if (eq_ok && r.proven) return ST_OK;
    return eq_ok ? ST_PENDING_EQ_HELD : ST_PENDING_EQ_FAILED;
}

JJS_HD uint32_t verify_item(const verify_params& P, uint64_t item, uint32_t* ws, bool write_c = true) {
    return finish_item(P, item, ws, prepare_item(P, item, write_c));
}

// prep_record <-> the device buffer: c at prep + 32 i, (a, |b|) at prep + 32 n + 32 i, flags at prep + 64 n + i
JJS_HD void store_prep(uint8_t* prep, uint64_t n, uint64_t i, const prep_record& r) {
    store_words(prep, i, r.c);
    words8 ab;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ab.w[k] = r.h.a.w[k]; ab.w[4 + k] = r.h.b.w[k]; }
    store_words(prep + 32 * n, i, ab);
    prep[64 * n + i] = (uint8_t)((r.malformed ? 1 : 0) | (r.valid ? 2 : 0) | (r.proven ? 4 : 0) | (r.h.b_neg ? 8 : 0));
}
JJS_HD prep_record load_prep(const uint8_t* prep, uint64_t n, uint64_t i) {
    prep_record r;
    const fe_src cs{prep, 32, 0}, abs_{prep + 32 * n, 32, 0};
    r.c = load_words(cs, i);
    const words8 ab = load_words(abs_, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.h.a.w[k] = ab.w[k]; r.h.b.w[k] = ab.w[4 + k]; }
    const uint8_t f = prep[64 * n + i];
    r.malformed = (f & 1) != 0; r.valid = (f & 2) != 0; r.proven = (f & 4) != 0; r.h.b_neg = (f & 8) != 0;
    return r;
}

// Resolve pass for an item the first pass left pending: every point that has not had its own subgroup
// test gets it now (`is_torsion_free`, src/keys/public.rs:159-164), then the reference's precedence applies.
// resolve_point is one such test (the j-th point without its own first-pass test; true beyond the last one),
// so that the device can give each point of an item to a different lane.
JJS_HD bool resolve_point(const verify_params& P, uint64_t item, uint32_t j) {
    // pick the source first, test once: lanes of a wave hold different j and must not take turns
    fe_src src = P.points[0];
    bool found = false;
    uint32_t seen = 0;
    const uint32_t tested = P.own_test_mask | (keyed_mode(P) ? P.key_points_mask : 0u);
    for (uint32_t k = 0; k < P.n_points; ++k) {
        if ((tested >> k) & 1u) continue;
        const bool hit = (seen++ == j);
        src.base = hit ? P.points[k].base : src.base;
        src.stride = hit ? P.points[k].stride : src.stride;
        src.off = hit ? P.points[k].off : src.off;
        found = found || hit;
    }
    fe_n pu = load_fq(src, item), pv = load_fq(src, item, 32);
    return is_torsion_free(pu, pv) || !found;
}
JJS_HD uint32_t resolve_status(bool all_torsion_free, bool eq_held) {
    return !all_torsion_free ? ST_INVALID_POINT : (eq_held ? ST_OK : ST_INVALID_SIGNATURE);
}
JJS_HD uint32_t resolve_item(const verify_params& P, uint64_t item, bool eq_held) {
    bool valid = true;
    for (uint32_t j = 0; j < P.n_points; ++j) valid = resolve_point(P, item, j) && valid;
    return resolve_status(valid, eq_held);
}

// ---- fixed-base comb table: entry (i, b) = b * 2^(COMB_BITS*i) * Base as an affine cached addend ----
// 1 / a: fq_inverse (fq_inv.h, division steps).  The power a^(q-2) it replaced stays as the cross-check of the tests and of
// tools/devcheck: ~300 products against ~11 k integer instructions.
JJS_HD fe_n fq_inverse_by_power(const fe_n& a) { return fq_pow_schedule(a, JJS_INV_SW, JJS_INV_SW_STEPS, JJS_INV_SW_TRAILING); }

JJS_HD void comb_entry_words(uint32_t* dst, const uint32_t (*base)[9], int i, int b) {
    fe_n bu = fq_as<1, 2>(fe_from_const<1, 1>(base[0])), bv = fq_as<1, 2>(fe_from_const<1, 1>(base[1]));
    ext_pt p = ext_from_affine(bu, bv);
    niels_pt n = to_niels(p);
    ext_pt acc = ext_identity();
    for (int bit = COMB_BITS * i + COMB_BITS - 1; bit >= 0; --bit) {
        acc = ext_double(acc, true);
        bool set = bit >= COMB_BITS * i && ((b >> (bit - COMB_BITS * i)) & 1);
        niels_pt addend = niels_select(set, n, niels_identity());
        acc = ext_add_niels(acc, addend, false, true);
    }
    fe_n zi = fq_inverse(acc.z);
    fe_n x = fq_mul(acc.x, zi), y = fq_mul(acc.y, zi);
    fe_n ypx = fq_reduce(fq_norm(fq_add(y, x)));
    fe_n ymx = fq_mul(fq_norm(fq_sub(y, x)), fq_one());      // times 1: same value, back below 2q
    fe_n t2d = fq_mul(fq_mul(x, y), fe_from_const<1, 1>(JJS_D2));
    for (int k = 0; k < 9; ++k) { dst[k] = ypx.l[k]; dst[9 + k] = ymx.l[k]; dst[18 + k] = t2d.l[k]; }
    dst[27] = 0;
}
JJS_HD void build_comb_entry(uint32_t* table, const uint32_t (*base)[9], int i, int b) {
    comb_entry_words(table + ((size_t)i * COMB_ENTRIES + b) * COMB_ENTRY_WORDS, base, i, b);
}

}  // namespace jjs
