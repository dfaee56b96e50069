// Key-table path: batches in which public keys (and per-key generators) repeat.
//
// The throughput path treats every public key as a fresh variable point: per signature it builds window tables
// for PK and R and runs 124 shared doublings (verify_core.h check_equation).  When a batch carries many
// signatures under the same key -- a validator set, an account signing many messages; SURVEY.md 8(d)'s workload
// has 4 096 keys under 2^20 signatures -- the key can be given what the generators G, G' have: a table of all its
// window multiples, built once per call, after which c*PK costs additions only.  Per call, on the device:
//
//   1. dedup      every key column (PK; PK' for the double scheme; PK and Gen for the per-item generator) is
//                 deduplicated by its 64 bytes in an open-addressing hash table (full byte comparison on a hash
//                 hit: two keys are merged only if they are the same bytes); keys get dense ids.
//   2. decision   if some column has more than n / KT_MIN_MULTIPLICITY distinct keys the tables would cost more
//                 than they save: a device flag sends the batch down the throughput path instead (both sets of
//                 kernels are queued, the one that is not wanted leaves at once; no host synchronisation).  The same
//                 happens when some key needed more than KT_MAX_PROBES probes (keys crafted to collide in the hash
//                 table must not be able to make the call slow) or when the key arena cannot be allocated.
//   3. per key    `is_valid` of the key point ONCE per key (src/keys/public.rs:159-164: canonical, on the curve,
//                 not the identity, torsion-free by the pairing test), the chain B_i = 2^(w i) * P and the tables
//                 {0 .. 2^(w-1)} * B_i for signed w-bit digits (w = 5 or 6, chosen with the decision).
//   4. per item   challenge hash as before (prepare_item in keyed mode: no half-size scalars, no combined test);
//                 then  acc = sum_i digit_i(c) * B_i  (51 or 43 additions, no doubling) + u*G from the fixed-base comb
//                 (16 additions), compared with R projectively: the reference's own equation u*G + c*PK == R
//                 (src/keys/public.rs:128-130), computed exactly by the complete addition law.
//   Subgroup membership of R: if the equation holds and the key is torsion-free then R = u*G + c*PK is in the
//   prime-order subgroup, and R != identity, on-curve are checked per item; if it fails, R's own test decides
//   between InvalidPoint and InvalidSignature in the resolve pass (as for the per-item generator scheme).
//
// ~330 k instructions per single verification instead of ~600 k, plus ~23 k wave-instructions per distinct key.
#pragma once
#include "verify_core.h"
#include "decode.h"
#include "ed29_quad.h"

namespace jjs {

// Signed digits of w bits: a scalar below 2^252 has kt_positions(w) of them (the top one unsigned and small), a table
// holds the multiples 0 .. 2^(w-1) of its base.  Wider windows trade per-key work and memory (positions x entries)
// for per-signature additions: 4 -> 64 additions / 83 KB per key, 5 -> 51 / 125 KB, 6 -> 43 / 204 KB.  The width is
// chosen on the device with the decision itself (key_spread_kernel): 6 bits when every column has at least
// KT_WIDE_MULTIPLICITY signatures per key (the 560 extra table additions per key are then repaid by the 8 saved per
// signature, with margin for the larger tables' cache footprint), else 5.  Measured on SURVEY.md 8(d)'s workload (256
// signatures per key; 2^20 single / double / var-gen, ms): 4 bits 11.01 / 19.82 / 14.75, 5 bits 11.00 / 19.06 / 13.98,
// 6 bits 10.81 / 18.59 / 13.26.
constexpr int KT_WINDOW_NARROW = 5, KT_WINDOW_WIDE = 6;
constexpr uint32_t KT_WIDE_MULTIPLICITY = 128;
JJS_HD constexpr int kt_positions(int w) { return (252 + w) / w; }
JJS_HD constexpr int kt_entries(int w) { return (1 << (w - 1)) + 1; }
JJS_HD constexpr int kt_table_words(int w) { return kt_entries(w) * ENTRY_WORDS; }
constexpr int KT_MAX_POSITIONS = kt_positions(KT_WINDOW_NARROW);
static_assert(KT_WINDOW_NARROW >= 4 && KT_WINDOW_WIDE <= 6 && KT_WINDOW_NARROW < KT_WINDOW_WIDE, "digit extraction assumes a digit spans at most two words");
constexpr uint32_t KT_MIN_MULTIPLICITY = 16;     // the tables pay from ~6 signatures per key; margin for their memory
constexpr uint32_t KT_MAX_PROBES = 128;          // hash-table probes per key before the batch gives up on key tables
constexpr uint32_t KT_KEY_MALFORMED = 1, KT_KEY_VALID = 2;
constexpr int KT_BASE_WORDS = 36;
// The bases and tables of the keys live in a pool that is sized by the number of distinct keys the slot's calls have
// carried, not by the batch size (verify_job.h setup_keys): a pool column holds `narrow` keys with narrow windows or `wide`
// keys with wide ones, and these are the words its two regions need for that.
JJS_HD constexpr size_t kt_max(size_t a, size_t b) { return a > b ? a : b; }
JJS_HD constexpr size_t kt_base_words_for_keys(size_t narrow, size_t wide) {
    return kt_max(narrow * kt_positions(KT_WINDOW_NARROW), wide * kt_positions(KT_WINDOW_WIDE)) * KT_BASE_WORDS;
}
JJS_HD constexpr size_t kt_table_words_for_keys(size_t narrow, size_t wide) {
    return kt_max(narrow * kt_positions(KT_WINDOW_NARROW) * kt_table_words(KT_WINDOW_NARROW),
                  wide * kt_positions(KT_WINDOW_WIDE) * kt_table_words(KT_WINDOW_WIDE));
}
// bytes of bases + tables of one key
JJS_HD constexpr size_t kt_key_bytes(int w) { return (size_t)kt_positions(w) * (KT_BASE_WORDS + kt_table_words(w)) * 4; }

struct key_column {
    fe_src src;              // the key points: 64 B affine (u || v); during the dedup of a wire call the 32 B encodings
    uint32_t* hash;          // open addressing, hash_mask + 1 slots: 0 = empty, else item index + 1
    uint32_t hash_mask;
    uint32_t key_bytes;      // bytes of src that identify a key: 64, or 32 while src is the compressed column
    uint32_t* rep;           // [n] first item found with the same key bytes
    uint32_t* keyid;         // [n] dense id of the item's key
    uint32_t* key_item;      // [max_keys] representative item of a key
    uint8_t* key_flags;      // [max_keys] KT_KEY_*
    uint8_t* key_undecodable;// [max_keys] wire calls: the key's encoding is not a point (decode.h)
    uint32_t* valid_ids;     // [max_keys] the ids of the keys whose point is valid, in the order the chain kernel found them (their
                             // number: key_params::counters[5 + column]): only these get window tables
    uint32_t* bases;         // [keys][positions][36]: 2^(w i) * P in extended coordinates (w = the window width of this batch)
    uint32_t* tables;        // [keys][positions][table words]: {0 .. 2^(w-1)} * base, cached-addend form
};
struct key_params {
    uint32_t n_cols;
    uint32_t max_keys;       // keys per column whose narrow-window tables fit the slot's table pool (also the bound of the ids
                             // that get a representative item); max_keys_wide: the same for wide windows
    uint32_t max_keys_wide;
    uint32_t quad_chains;    // the chain of a key on four lanes (kt_chain_key_quad) instead of one: where the batch waits for the chains
    uint64_t seed;           // per-call seed of the dedup hash (kt_hash)
    key_column col[2];
    uint32_t* counters;      // [c] distinct keys of column c; [2] the decision: 0 = throughput path, else the window width of
                             // the key-table path; [3] a probe sequence overflowed; [4] the keys repeat but their tables do
                             // not fit the pool (the host grows it for the next call); [5 + c] valid keys of column c
    uint32_t force_window;   // profiling build only (0 in the product): KT_WINDOW_NARROW = never the wide windows
    uint32_t keep_order;     // profiling build only (0 in the product): the lanes take the items in the caller's order
    uint64_t n;
    // the items grouped by their key of column 0 (counting sort on the device): lane i of key_verify_kernel takes item
    // order[i], so the lanes of a wave look up the tables of one or two keys instead of 64
    uint32_t* order;         // [n]
    uint32_t* key_cursor;    // items per key, then the start of each key's run (advanced while scattering); dense, or a 64-byte
                             // line per key when the batch has few keys (device_kernels.h cursor_stride)
};

// column `idx` (0 or 1) of K, chosen field by field: indexing the kernel-argument struct with a run-time index
// would make the compiler copy it to scratch memory
JJS_HD key_column kt_col(const key_params& K, int32_t idx) {
    const key_column &a = K.col[0], &b = K.col[1];
    const bool z = idx == 0;
    key_column c;
    c.src.base = z ? a.src.base : b.src.base; c.src.stride = z ? a.src.stride : b.src.stride; c.src.off = z ? a.src.off : b.src.off;
    c.hash = z ? a.hash : b.hash; c.hash_mask = z ? a.hash_mask : b.hash_mask; c.key_bytes = z ? a.key_bytes : b.key_bytes;
    c.rep = z ? a.rep : b.rep; c.keyid = z ? a.keyid : b.keyid; c.key_item = z ? a.key_item : b.key_item;
    c.key_flags = z ? a.key_flags : b.key_flags; c.key_undecodable = z ? a.key_undecodable : b.key_undecodable; c.bases = z ? a.bases : b.bases; c.tables = z ? a.tables : b.tables;
    c.valid_ids = z ? a.valid_ids : b.valid_ids;
    return c;
}

// seed: drawn by the host for every call (key_params::seed), so that keys which share a probe sequence cannot be computed
// ahead of time; the probe limit above stays as the backstop
JJS_HD uint64_t kt_hash(const fe_src& src, uint64_t item, uint32_t bytes, uint64_t seed) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ seed;
#pragma unroll
    for (uint32_t off = 0; off < 64; off += 32) {
        if (off >= bytes) break;
        const words8 w = load_words(src, item, off);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            h ^= ((uint64_t)w.w[i + 1] << 32) | w.w[i];
            h *= 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
    }
    return h;
}
JJS_HD bool kt_same_key(const fe_src& src, uint64_t a, uint64_t b, uint32_t bytes) {
    uint32_t diff = 0;
#pragma unroll
    for (uint32_t off = 0; off < 64; off += 32) {
        if (off >= bytes) break;
        const words8 x = load_words(src, a, off), y = load_words(src, b, off);
#pragma unroll
        for (int i = 0; i < 8; ++i) diff |= x.w[i] ^ y.w[i];
    }
    return diff == 0;
}

JJS_HD uint32_t* kt_base(const key_column& C, uint32_t id, uint32_t pos, int w) {
    return C.bases + ((size_t)id * kt_positions(w) + pos) * KT_BASE_WORDS;
}
JJS_HD uint32_t* kt_table(const key_column& C, uint32_t id, uint32_t pos, int w) {
    return C.tables + ((size_t)id * kt_positions(w) + pos) * kt_table_words(w);
}
JJS_HD void kt_store_ext(uint32_t* dst, const ext_pt& p) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { dst[i] = p.x.l[i]; dst[9 + i] = p.y.l[i]; dst[18 + i] = p.z.l[i]; dst[27 + i] = p.t.l[i]; }
}
JJS_HD ext_pt kt_load_ext(const uint32_t* src) {
    ext_pt p;
#pragma unroll
    for (int i = 0; i < 9; ++i) { p.x.l[i] = src[i]; p.y.l[i] = src[9 + i]; p.z.l[i] = src[18 + i]; p.t.l[i] = src[27 + i]; }
    return p;
}

// Wire calls (decode.h): the key columns arrive as 32-byte encodings and are deduplicated as such; a key is then
// decompressed ONCE (one square root per key instead of one per signature) into the affine column at its
// representative item, and every other item of the key copies the 64 bytes from there.  `out` is the decoded
// column (n x 64), `bad` the per-item malformed flags of the call.
JJS_HD void kt_decode_key(const key_column& C, uint32_t id, uint8_t* out, const dlog_tables& T) {
    const uint64_t item = C.key_item[id];
    const decoded_point d = decompress_point(load_words(C.src, item), T);
    store_words(out, 2 * item, d.u);
    store_words(out, 2 * item + 1, d.v);
    C.key_undecodable[id] = d.ok ? 0 : 1;
}
JJS_HD void kt_unpack_item(const key_column& C, uint64_t item, uint8_t* out, uint8_t* bad) {
    const uint32_t r = C.rep[item];
    if (r != (uint32_t)item) {
        const fe_src o{out, 64, 0};
        store_words(out, 2 * item, load_words(o, r));
        store_words(out, 2 * item + 1, load_words(o, r, 32));
    }
    if (C.key_undecodable[C.keyid[item]]) bad[item] = 1;
}

// one key: `is_valid` of its point (returned), and the chain of bases
JJS_HD bool kt_chain_key(const key_column& C, uint32_t id, int w) {
    const uint64_t item = C.key_item[id];
    const words8 uw = load_words(C.src, item), vw = load_words(C.src, item, 32);
    const bool canonical = words_lt(uw, JJS_Q_WORDS) && words_lt(vw, JJS_Q_WORDS);
    const fe_n pu = fq_from_words(uw), pv = fq_from_words(vw);
    const bool valid = canonical && point_on_curve_not_identity(pu, pv) && is_torsion_free(pu, pv);
    C.key_flags[id] = (uint8_t)((canonical ? 0u : KT_KEY_MALFORMED) | (valid ? KT_KEY_VALID : 0u));
    ext_pt p = ext_from_affine(pu, pv);
    kt_store_ext(kt_base(C, id, 0, w), p);
    const uint32_t positions = (uint32_t)kt_positions(w);
    for (uint32_t pos = 1; pos < positions; ++pos) {
#pragma unroll 1
        for (int j = 0; j < w; ++j) p = ext_double(p, j == w - 1);
        kt_store_ext(kt_base(C, id, pos, w), p);
    }
    return valid;
}

#if defined(__HIPCC__)
// The chain of bases of a key on FOUR adjacent lanes (ed29_quad.h ext_double_quad): the chain is 252 doublings long whatever
// the batch, and batches of up to 2^18 items wait for it.
// the chain of bases of one key on a quad (kt_chain_key without the validity test, which other lanes run: kt_key_flags)
__device__ __forceinline__ void kt_chain_key_quad(const key_column& C, uint32_t id, int w, uint32_t j) {
    const uint64_t item = C.key_item[id];
    const fe_n pu = fq_from_words(load_words(C.src, item)), pv = fq_from_words(load_words(C.src, item, 32));
    ext_pt p = ext_from_affine(pu, pv);
    if (j == 0) kt_store_ext(kt_base(C, id, 0, w), p);
    const uint32_t positions = (uint32_t)kt_positions(w);
    for (uint32_t pos = 1; pos < positions; ++pos) {
        fe_n own;
#pragma unroll 1
        for (int k = 0; k < w; ++k) p = ext_double_quad(p, j, own);
        uint32_t* dst = kt_base(C, id, pos, w) + 9 * j;           // lane j holds coordinate j of the new base
#pragma unroll
        for (int i = 0; i < 9; ++i) dst[i] = own.l[i];
    }
}
#endif
// `is_valid` of one key's point (the other half of kt_chain_key)
JJS_HD bool kt_key_flags(const key_column& C, uint32_t id) {
    const uint64_t item = C.key_item[id];
    const words8 uw = load_words(C.src, item), vw = load_words(C.src, item, 32);
    const bool canonical = words_lt(uw, JJS_Q_WORDS) && words_lt(vw, JJS_Q_WORDS);
    const fe_n pu = fq_from_words(uw), pv = fq_from_words(vw);
    const bool valid = canonical && point_on_curve_not_identity(pu, pv) && is_torsion_free(pu, pv);
    C.key_flags[id] = (uint8_t)((canonical ? 0u : KT_KEY_MALFORMED) | (valid ? KT_KEY_VALID : 0u));
    return valid;
}

// table[j] = j * P for j = 0 .. entries-1, P in extended coordinates (the cached-addend entries keep their Z)
JJS_HD void kt_build_table(uint32_t* tab, const ext_pt& p1, int entries) {
    const niels_pt n1 = to_niels(p1);
    store_niels(tab, niels_identity());
    store_niels(tab + ENTRY_WORDS, n1);
    ext_pt acc = ext_double(p1, true);
    store_niels(tab + 2 * ENTRY_WORDS, to_niels(acc));
    for (int j = 3; j < entries; ++j) {
        acc = ext_add_niels(acc, n1, false, true);
        store_niels(tab + j * ENTRY_WORDS, to_niels(acc));
    }
}
JJS_HD void kt_table_lane(const key_column& C, uint32_t id, uint32_t pos, int w) {
    kt_build_table(kt_table(C, id, pos, w), kt_load_ext(kt_base(C, id, pos, w)), kt_entries(w));
}

// s + sum_{i < positions-1} 2^(w-1) * 2^(w i): digit i of the sum, minus 2^(w-1), is signed digit i of s; the top
// digit is unsigned (at most 5 for s < 2^252)
JJS_HD constexpr uint32_t kt_recode_word(int w, int word) {
    uint32_t v = 0;
    for (int i = 0; i < kt_positions(w) - 1; ++i) {
        const int bit = w * i + w - 1;
        if ((bit >> 5) == word) v |= 1u << (bit & 31);
    }
    return v;
}
JJS_HD words8 kt_recode(const words8& s, int w) {
    constexpr uint32_t KN[8] = {kt_recode_word(KT_WINDOW_NARROW, 0), kt_recode_word(KT_WINDOW_NARROW, 1), kt_recode_word(KT_WINDOW_NARROW, 2),
                                kt_recode_word(KT_WINDOW_NARROW, 3), kt_recode_word(KT_WINDOW_NARROW, 4), kt_recode_word(KT_WINDOW_NARROW, 5),
                                kt_recode_word(KT_WINDOW_NARROW, 6), kt_recode_word(KT_WINDOW_NARROW, 7)};
    constexpr uint32_t KW[8] = {kt_recode_word(KT_WINDOW_WIDE, 0), kt_recode_word(KT_WINDOW_WIDE, 1), kt_recode_word(KT_WINDOW_WIDE, 2),
                                kt_recode_word(KT_WINDOW_WIDE, 3), kt_recode_word(KT_WINDOW_WIDE, 4), kt_recode_word(KT_WINDOW_WIDE, 5),
                                kt_recode_word(KT_WINDOW_WIDE, 6), kt_recode_word(KT_WINDOW_WIDE, 7)};
    const bool wide = w == KT_WINDOW_WIDE;
    words8 r;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t t = (uint64_t)s.w[i] + (wide ? KW[i] : KN[i]) + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return r;
}
// acc + (digit `pos` of the recoded scalar) * base_pos; pos and w are wave-uniform, the digit is per lane
JJS_HD ext_pt kt_add_digit(const ext_pt& acc, const uint32_t* tab, const words8& sc, int pos, int w) {
    const int bit = w * pos, wi = bit >> 5, sh = bit & 31;
    uint32_t v = word_at(sc, wi) >> sh;
    if (sh + w > 32) v |= word_at(sc, wi + 1 > 7 ? 7 : wi + 1) << (32 - sh);
    const uint32_t raw = v & ((1u << w) - 1u);
    const bool top = pos == kt_positions(w) - 1;
    const int d = top ? (int)(word_at(sc, wi) >> sh) : (int)raw - (1 << (w - 1));
    const bool neg = d < 0;
    uint32_t idx = (uint32_t)(neg ? -d : d);
    idx = idx >= (uint32_t)kt_entries(w) ? (uint32_t)kt_entries(w) - 1u : idx;     // out-of-range (malformed) scalars only
    return ext_add_niels(acc, load_niels(tab + idx * ENTRY_WORDS), neg, true);
}
// acc + s * P for the key `id` of column C, s < 2^252: one addition per position, no doubling.  T of the result is valid.
JJS_HD ext_pt kt_add_scalar(ext_pt acc, const key_column& C, uint32_t id, const words8& s, int w) {
    const words8 sc = kt_recode(s, w);
    for (int pos = kt_positions(w) - 1; pos >= 0; --pos) acc = kt_add_digit(acc, kt_table(C, id, (uint32_t)pos, w), sc, pos, w);
    return acc;
}

// The equations of one item through the key tables (step 4 above).  A key whose point is not valid has no tables (the table
// kernel builds them for the valid keys only: an invalid key makes every one of its items InvalidPoint whatever the equation
// says, and SURVEY.md 8(d)'s mix carries ~2 600 such keys beside its 4 096 good ones): its items add whatever the pool
// holds there and their verdict ignores the sum.  r = prepare_item's record in keyed mode:
// challenge, malformed (u, m, the R points), valid (every R point on the curve and not the identity).
JJS_HD uint32_t kt_finish_item(const verify_params& P, const key_params& K, uint64_t item, const prep_record& r) {
    const int w = (int)*P.key_flag;                  // the window width this batch's tables were built with
    const words8 u = load_words(P.u, item);
    bool keys_valid = true, keys_malformed = false, eq_ok = true;
    for (uint32_t e = 0; e < P.n_eq; ++e) {
        const eq_desc& E = P.eq[e];
        ext_pt acc = ext_identity();
        {
            const key_column C = kt_col(K, E.pk_col);
            const uint32_t id = C.keyid[item], f = C.key_flags[id];
            keys_valid = keys_valid && (f & KT_KEY_VALID) != 0;
            keys_malformed = keys_malformed || (f & KT_KEY_MALFORMED) != 0;
            acc = kt_add_scalar(acc, C, id, r.c, w);                          // c * PK
        }
        if (E.comb) {
            // JJS_SKIP bit 4 (profiling build only, constant false in the product): 256 entries per comb row, a
            // cache-resident working set -- what the comb gathers cost
            acc = add_comb(acc, E.comb, u, JJS_SKIP(P, 16u));                 // + u * G (G')
        } else {
            const key_column C = kt_col(K, E.gen_col);
            const uint32_t id = C.keyid[item], f = C.key_flags[id];
            keys_valid = keys_valid && (f & KT_KEY_VALID) != 0;
            keys_malformed = keys_malformed || (f & KT_KEY_MALFORMED) != 0;
            acc = kt_add_scalar(acc, C, id, u, w);                             // + u * Gen
        }
        const fe_n ru = load_fq(E.r, item), rv = load_fq(E.r, item, 32);
        eq_ok = ext_eq_affine(acc, ru, rv) && eq_ok;
    }
    // u < r is checked here: the head launch of prepare_kernel, which made r, does not read u (verify_core.h prep_phase)
    if (r.malformed || keys_malformed || !words_lt(u, JJS_FR_WORDS)) return ST_MALFORMED;
    if (!r.valid || !keys_valid) return ST_INVALID_POINT;
    if (JJS_SKIP(P, 1u)) return eq_ok ? ST_OK : ST_INVALID_SIGNATURE;        // profiling only: no resolve pass
    // the equations hold: every R is a sum of torsion-free points; they fail: R's own subgroup test decides
    return eq_ok ? ST_OK : ST_PENDING_EQ_FAILED;
}

}  // namespace jjs
