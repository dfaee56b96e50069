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
//   * c*PK: signed 4-bit fixed windows over a per-lane table {0..8}*PK kept in a global-memory
//     workspace (144 B per entry, one lane's table contiguous); 252 shared doublings.
//   * u*G (and u*G'): 8-bit fixed-base comb, 32 mixed additions from a 917 KB table (L2-resident).
//   * var-gen: u*Gen joins the same window loop with a second per-lane table (Straus).
//   * subgroup check: order-8 Tate pairing residue test, one exponentiation per point instead of [r]P.
#pragma once
#include "ed29.h"
#include "hades29.h"

namespace jjs {

enum : uint32_t { ST_OK = 0, ST_INVALID_POINT = 1, ST_INVALID_SIGNATURE = 2, ST_MALFORMED = 3 };

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
};
struct verify_params {
    uint32_t n_hash, n_points, n_eq, pad_;
    fe_src hash_in[10];
    fe_src points[4];
    eq_desc eq[2];
    fe_src u;
    uint64_t n;
    uint8_t* status;                 // n bytes, or nullptr
    unsigned long long* tally;       // 4 counters, or nullptr
    uint8_t* c_out;                  // n x 32 bytes challenge (debug export), or nullptr
    uint32_t* workspace;             // WS_WORDS_PER_LANE words per resident lane
};

constexpr int TABLE_ENTRIES = 9;                 // {0..8} * P
constexpr int ENTRY_WORDS = 36;                  // 4 coordinates x 9 limbs
constexpr int TABLE_WORDS = TABLE_ENTRIES * ENTRY_WORDS;
constexpr int WS_WORDS_PER_LANE = 2 * TABLE_WORDS;
constexpr int COMB_WINDOWS = 32, COMB_ENTRIES = 256;
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
        acc = ext_add_niels(acc, n1, false, true);
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
JJS_HD ext_pt add_window(const ext_pt& acc, const uint32_t* tab, const words8& sc, int w, bool need_t) {
    uint32_t nib = nibble(sc, w);
    int d = (w == 63) ? (int)nib : (int)nib - 8;
    bool neg = d < 0;
    uint32_t idx = (uint32_t)(neg ? -d : d);
    return ext_add_niels(acc, load_niels(tab + idx * ENTRY_WORDS), neg, need_t);
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
JJS_HD bool is_torsion_free(const fe_n& u, const fe_n& v) {
    auto opv = fq_add(fq_one(), v);                       // 1 + v          <2,3>
    auto omv = fq_sub(fq_one(), v);                       // 1 - v          <3,4>
    fe_n w = fq_mul(omv, u);                              // W = (1-v) u
    fe_n x = fq_mul(opv, u);                              // X = (1+v) u
    fe_n xw[2] = {x, w};
    auto l1 = fq_norm(fq_add(opv, fq_dot_const<2, 2>(JJS_PAIR_L1, xw)));                       // <1,5>
    auto l2 = fq_norm(fq_add(opv, fq_mul(x, fe_from_const<1, 1>(JJS_PAIR_NEG_L2))));          // <1,5>
    auto vv = fq_norm(fq_sub(x, w));                                                           // <1,5>
    fe_n a = fq_mul(fq_sqr(l1), l2);
    a = fq_sqr(fq_mul(a, fq_sqr(vv)));                    // L1^4 L2^2 V^4
    fe_n y = fq_mul(fq_mul(x, w), fe_from_const<1, 1>(JJS_PAIR_B));
    fe_n y2 = fq_sqr(y);
    fe_n y7 = fq_mul(fq_mul(fq_sqr(y2), y2), y);
    fe_n g = fq_mul(a, y7);
    fe_n e = fq_pow_public(g, JJS_PAIR_EXP_WORDS, 252);
    return fq_eq(e, fq_one());
}

// is_torsion_free && is_on_curve && !is_identity  (src/keys/public.rs:159-164)
JJS_HD bool point_is_valid(const fe_n& u, const fe_n& v) {
    bool on = affine_on_curve(u, v);
    bool id = affine_is_identity(u, v);
    bool tf = is_torsion_free(u, v);
    return tf && on && !id;
}

// u*Gen + c*PK == R
JJS_HD bool check_equation(const eq_desc& E, uint64_t item, uint32_t* ws, const words8& u, const words8& c) {
    uint32_t* tab_pk = ws;
    uint32_t* tab_gen = ws + TABLE_WORDS;
    {
        fe_n pu = load_fq(E.pk, item), pv = load_fq(E.pk, item, 32);
        build_point_table(tab_pk, pu, pv);
    }
    const bool varbase = (E.comb == nullptr);
    words8 su;
    if (varbase) {
        fe_n gu = load_fq(E.gen, item), gv = load_fq(E.gen, item, 32);
        build_point_table(tab_gen, gu, gv);
        su = recode_signed4(u);
    }
    const words8 sc = recode_signed4(c);
    ext_pt acc = ext_identity();
    for (int w = 63; w >= 0; --w) {
        if (w != 63) {
            acc = ext_double(acc, false);
            acc = ext_double(acc, false);
            acc = ext_double(acc, false);
            acc = ext_double(acc, true);
        }
        if (varbase) {
            acc = add_window(acc, tab_pk, sc, w, true);
            acc = add_window(acc, tab_gen, su, w, false);
        } else {
            acc = add_window(acc, tab_pk, sc, w, w == 0);
        }
    }
    if (!varbase) {
        for (int i = 0; i < COMB_WINDOWS; ++i) {
            uint32_t byte = (word_at(u, i >> 2) >> ((i & 3) * 8)) & 255u;
            const u32x4* p = reinterpret_cast<const u32x4*>(E.comb + ((size_t)i * COMB_ENTRIES + byte) * COMB_ENTRY_WORDS);
            uint32_t w[COMB_ENTRY_WORDS];
#pragma unroll
            for (int k = 0; k < COMB_ENTRY_WORDS / 4; ++k) { u32x4 v = p[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
            fe_t ypx, ymx, t2d;
#pragma unroll
            for (int k = 0; k < 9; ++k) { ypx.l[k] = w[k]; ymx.l[k] = w[9 + k]; t2d.l[k] = w[18 + k]; }
            acc = ext_add_affine_niels(acc, ypx, ymx, t2d, i != COMB_WINDOWS - 1);
        }
    }
    fe_n ru = load_fq(E.r, item), rv = load_fq(E.r, item, 32);
    return ext_eq_affine(acc, ru, rv);
}

// one signature; ws = this lane's WS_WORDS_PER_LANE workspace words
JJS_HD uint32_t verify_item(const verify_params& P, uint64_t item, uint32_t* ws, bool write_c = true) {
    // 1. encodings: every transcript element (all point coordinates and m) < q, u < r
    const words8 u = load_words(P.u, item);
    bool malformed = !words_lt(u, JJS_FR_WORDS);
    for (uint32_t e = 0; e < P.n_hash; ++e) malformed = malformed || !words_lt(load_words(P.hash_in[e], item), JJS_Q_WORDS);

    // 2. point validity (InvalidPoint takes precedence over InvalidSignature)
    bool valid = true;
    for (uint32_t k = 0; k < P.n_points; ++k) {
        fe_n pu = load_fq(P.points[k], item), pv = load_fq(P.points[k], item, 32);
        valid = point_is_valid(pu, pv) && valid;
    }

    // 3. challenge
    fe_n digest = poseidon_digest((int)P.n_hash, [&](int e) { return load_fq(P.hash_in[e], item); });
    const words8 c = truncate250(digest);
    if (P.c_out && write_c) store_words(P.c_out, item, c);

    // 4. equations
    bool eq_ok = true;
    for (uint32_t k = 0; k < P.n_eq; ++k) eq_ok = check_equation(P.eq[k], item, ws, u, c) && eq_ok;

    return malformed ? ST_MALFORMED : (!valid ? ST_INVALID_POINT : (!eq_ok ? ST_INVALID_SIGNATURE : ST_OK));
}

// ---- fixed-base comb table: entry (i, b) = b * 256^i * Base as an affine cached addend ------------
JJS_HD fe_n fq_inverse(const fe_n& a) { return fq_pow_public(a, JJS_QM2_WORDS, 255); }

JJS_HD void build_comb_entry(uint32_t* table, const uint32_t (*base)[9], int i, int b) {
    fe_n bu = fq_as<1, 2>(fe_from_const<1, 1>(base[0])), bv = fq_as<1, 2>(fe_from_const<1, 1>(base[1]));
    ext_pt p = ext_from_affine(bu, bv);
    niels_pt n = to_niels(p);
    ext_pt acc = ext_identity();
    for (int bit = 8 * i + 7; bit >= 0; --bit) {
        acc = ext_double(acc, true);
        bool set = bit >= 8 * i && ((b >> (bit - 8 * i)) & 1);
        niels_pt addend = niels_select(set, n, niels_identity());
        acc = ext_add_niels(acc, addend, false, true);
    }
    fe_n zi = fq_inverse(acc.z);
    fe_n x = fq_mul(acc.x, zi), y = fq_mul(acc.y, zi);
    fe_n ypx = fq_reduce(fq_norm(fq_add(y, x)));
    fe_n ymx = fq_mul(fq_norm(fq_sub(y, x)), fq_one());      // times 1: same value, back below 2q
    fe_n t2d = fq_mul(fq_mul(x, y), fe_from_const<1, 1>(JJS_D2));
    uint32_t* dst = table + ((size_t)i * COMB_ENTRIES + b) * COMB_ENTRY_WORDS;
    for (int k = 0; k < 9; ++k) { dst[k] = ypx.l[k]; dst[9 + k] = ymx.l[k]; dst[18 + k] = t2d.l[k]; }
    dst[27] = 0;
}

}  // namespace jjs
