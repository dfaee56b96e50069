// Batch multisignature share verification and combination: many transcripts at once,
// transcript t owning participants [offsets[t], offsets[t+1]) of the flattened arrays.  Mirrors
//   multisig_common          /root/reference/src/multisig.rs:440-500  (d_i, aggregate key, a, RSa, c)
//   delinearization_coeff    src/multisig.rs:393-409
//   verify_share             src/multisig.rs:284-309 / verify_share_with_coefficients :366-387
//                            z_i*G + (c*d_i)*PK_i == R_i + a*S_i
//   combine                  src/multisig.rs:326-360  (u = sum z_i, R = RSa)
//   aggregate_pk             src/multisig.rs:154-156
// Like the reference these functions do not validate the points (they take JubJubExtended values);
// only the encodings are checked (status 3).  Seven passes, each one lane per participant or per
// transcript; the hash chains (2 + 2n and 3 + 4n inputs) run inside a lane -- or, when the call has few lanes (a long
// transcript or two: the sponge is one chain of (3 + 4n) / 4 permutations), on eight lanes each (msig_params::hash_lanes).
#pragma once
#include "sign_core.h"
#include "safe_tag.h"

namespace jjs {

enum : uint32_t { ST_INVALID_SHARE = 4, ST_INVALID_TRANSCRIPT = 5 };
constexpr int EXT_WORDS = 36;

struct msig_params {
    const uint8_t *z, *PK, *R, *S, *m;   // z: N x 32; PK, R, S: N x 64 affine; m: B x 32
    const uint32_t* offsets;             // B + 1 (device)
    uint32_t n_transcripts, pad_;
    uint64_t n_total;
    uint8_t* share_status;               // N bytes: 0 ok, 3 malformed encoding, 4 invalid share
    uint8_t *agg_pk, *sig_u, *sig_R;     // B x 64, B x 32, B x 64 (what aggregate_pk / combine return)
    uint8_t* transcript_status;          // B bytes: 0 = combine returns the signature; else the first share's failure
    uint32_t *tr_of, *d_words, *dpk, *e_pt, *a_words, *c_words;   // scratch: N, N x 8, N x 36, N x 36, B x 8, B x 8
    const uint32_t* tags;                // SAFE tags [JJS_LONG_TAGS][9]: transcripts of up to JJS_MSIG_MAX_PARTICIPANTS participants
    uint32_t* long_tags;                 // scratch [B][2][9]: the two tags of every longer transcript, computed by pass 0
                                         // (csrc/safe_tag.h); rows of the other transcripts are neither written nor read
    uint32_t max_table_participants;
    uint32_t hash_lanes;                 // 1, or 8 (device, passes 1, 2 and 4 of a call with few lanes: hades_permute's coop mode -- eight
                                         // adjacent lanes run the item together, sharing its hash; everything else they do eight times over,
                                         // every lane storing the same values)
    const uint32_t* comb_g;
    uint32_t* lane_ws;                   // WS_WORDS_PER_LANE per resident lane
};

JJS_HD fe_n load_tag(const uint32_t* tags, int n_inputs) {
    fe_n t;
    for (int i = 0; i < 9; ++i) t.l[i] = tags[(size_t)n_inputs * 9 + i];
    return t;
}
// the tag of transcript t's hash number `which` (0: delinearisation, 2 + 2n inputs; 1: a, 3 + 4n inputs)
JJS_HD fe_n msig_tag(const msig_params& P, uint32_t t, uint32_t participants, int which, int n_inputs) {
    if (participants > P.max_table_participants) return load_tag(P.long_tags, (int)(2 * t) + which);
    return load_tag(P.tags, n_inputs);
}
JJS_HD void store_ext(uint32_t* dst, const ext_pt& p) {
    for (int i = 0; i < 9; ++i) { dst[i] = p.x.l[i]; dst[9 + i] = p.y.l[i]; dst[18 + i] = p.z.l[i]; dst[27 + i] = p.t.l[i]; }
}
JJS_HD ext_pt load_ext(const uint32_t* src) {
    ext_pt p;
    for (int i = 0; i < 9; ++i) { p.x.l[i] = src[i]; p.y.l[i] = src[9 + i]; p.z.l[i] = src[18 + i]; p.t.l[i] = src[27 + i]; }
    return p;
}
JJS_HD words8 load_w8(const uint32_t* p) { words8 w; for (int i = 0; i < 8; ++i) w.w[i] = p[i]; return w; }
JJS_HD void store_w8(uint32_t* p, const words8& w) { for (int i = 0; i < 8; ++i) p[i] = w.w[i]; }

// pass 0 (lane per transcript): participant -> transcript map; the SAFE tags of a transcript beyond the generated table
JJS_HD void msig_map_item(const msig_params& P, uint32_t t) {
    for (uint32_t i = P.offsets[t]; i < P.offsets[t + 1]; ++i) P.tr_of[i] = t;
    const uint32_t cnt = P.offsets[t + 1] - P.offsets[t];
    if (cnt > P.max_table_participants) {
        uint32_t q[8], tag[9];
        for (int k = 0; k < 8; ++k) q[k] = JJS_Q_WORDS[k];
        safe_tag_limbs(2u + 2u * cnt, q, tag);
        for (int k = 0; k < 9; ++k) P.long_tags[18 * (size_t)t + k] = tag[k];
        safe_tag_limbs(3u + 4u * cnt, q, tag);
        for (int k = 0; k < 9; ++k) P.long_tags[18 * (size_t)t + 9 + k] = tag[k];
    }
}
// pass 1 (lane per participant): d_i = H(pk_i, pk_lo .. pk_hi), D_i = d_i * PK_i
JJS_HD void msig_delin_item(const msig_params& P, uint64_t i, uint32_t* ws, int coop = -1) {
    const uint32_t t = P.tr_of[i], lo = P.offsets[t], hi = P.offsets[t + 1];
    const int n_in = 2 + 2 * (int)(hi - lo);
    const fe_src pk{P.PK, 64, 0};
    fe_n dg = poseidon_digest_tagged(n_in, msig_tag(P, t, hi - lo, 0, n_in), [&](int e) {
        return e < 2 ? load_fq(pk, i, 32u * (uint32_t)e) : load_fq(pk, lo + (uint64_t)((e - 2) >> 1), 32u * (uint32_t)(e & 1));
    }, coop);
    const words8 d = truncate250(dg);
    store_w8(P.d_words + 8 * i, d);
    build_point_table(ws, load_fq(pk, i), load_fq(pk, i, 32));
    store_ext(P.dpk + EXT_WORDS * i, table_mul(ws, d, true));
}
JJS_HD affine_words sum_points_affine(const uint32_t* pts, uint32_t lo, uint32_t hi) {
    ext_pt acc = ext_identity();
    for (uint32_t i = lo; i < hi; ++i) acc = ext_add_niels(acc, to_niels(load_ext(pts + (size_t)EXT_WORDS * i)), false, true);
    return to_affine_words(acc);
}
// pass 2 (lane per transcript): pk_agg = sum D_i;  a = H(pk_agg, m, R_lo, S_lo, ...)
JJS_HD void msig_agg_item(const msig_params& P, uint32_t t, int coop = -1) {
    const uint32_t lo = P.offsets[t], hi = P.offsets[t + 1];
    const affine_words agg = sum_points_affine(P.dpk, lo, hi);
    store_point(P.agg_pk, t, agg);
    const int n_in = 3 + 4 * (int)(hi - lo);
    const fe_src aggs{P.agg_pk, 64, 0}, ms{P.m, 32, 0}, rs{P.R, 64, 0}, ss{P.S, 64, 0};
    fe_n dg = poseidon_digest_tagged(n_in, msig_tag(P, t, hi - lo, 1, n_in), [&](int e) {
        if (e < 2) return load_fq(aggs, t, 32u * (uint32_t)e);
        if (e == 2) return load_fq(ms, t);
        const int k = e - 3;                     // R_i.u, R_i.v, S_i.u, S_i.v per participant
        const uint64_t idx = lo + (uint64_t)(k >> 2);
        return (k & 2) ? load_fq(ss, idx, 32u * (uint32_t)(k & 1)) : load_fq(rs, idx, 32u * (uint32_t)(k & 1));
    }, coop);
    store_w8(P.a_words + 8 * t, truncate250(dg));
}
// pass 3 (lane per participant): E_i = R_i + a * S_i
JJS_HD void msig_commit_item(const msig_params& P, uint64_t i, uint32_t* ws) {
    const uint32_t t = P.tr_of[i];
    const fe_src rs{P.R, 64, 0}, ss{P.S, 64, 0};
    build_point_table(ws, load_fq(ss, i), load_fq(ss, i, 32));
    ext_pt as = table_mul(ws, load_w8(P.a_words + 8 * t), true);
    ext_pt r = ext_from_affine(load_fq(rs, i), load_fq(rs, i, 32));
    store_ext(P.e_pt + EXT_WORDS * i, ext_add_niels(as, to_niels(r), false, true));
}
// pass 4 (lane per transcript): RSa = sum E_i, c = H(RSa, pk_agg, m), u = sum z_i
JJS_HD void msig_final_item(const msig_params& P, uint32_t t, int coop = -1) {
    const uint32_t lo = P.offsets[t], hi = P.offsets[t + 1];
    const affine_words rsa = sum_points_affine(P.e_pt, lo, hi);
    store_point(P.sig_R, t, rsa);
    const fe_src sr{P.sig_R, 64, 0}, aggs{P.agg_pk, 64, 0}, ms{P.m, 32, 0}, zs{P.z, 32, 0};
    fe_n dg = poseidon_digest(5, [&](int e) {
        return e < 2 ? load_fq(sr, t, 32u * (uint32_t)e) : (e < 4 ? load_fq(aggs, t, 32u * (uint32_t)(e - 2)) : load_fq(ms, t));
    }, coop);
    store_w8(P.c_words + 8 * t, truncate250(dg));
    // u = sum z_i mod r
    words8 u = small_words(0);
    for (uint32_t i = lo; i < hi; ++i) {
        const words8 z = load_words(zs, i);
        uint64_t carry = 0;
        words8 s, d;
        for (int k = 0; k < 8; ++k) { uint64_t x = (uint64_t)u.w[k] + z.w[k] + carry; s.w[k] = (uint32_t)x; carry = x >> 32; }
        uint32_t borrow = 0;
        for (int k = 0; k < 8; ++k) { uint64_t x = (uint64_t)s.w[k] - JJS_FR_WORDS[k] - borrow; d.w[k] = (uint32_t)x; borrow = (uint32_t)(x >> 63); }
        u = select_words(borrow != 0, s, d);      // both addends < r < 2^252: no carry out of 256 bits
    }
    store_words(P.sig_u, t, u);
}
// pass 5 (lane per participant): z_i*G + (c*d_i)*PK_i == E_i
JJS_HD void msig_share_item(const msig_params& P, uint64_t i, uint32_t* ws) {
    const uint32_t t = P.tr_of[i];
    const fe_src pk{P.PK, 64, 0}, zs{P.z, 32, 0}, rs{P.R, 64, 0}, ss{P.S, 64, 0}, ms{P.m, 32, 0};
    const words8 z = load_words(zs, i);
    bool malformed = !words_lt(z, JJS_FR_WORDS) || !words_lt(load_words(ms, t), JJS_Q_WORDS);
    for (int e = 0; e < 2; ++e) {
        malformed = malformed || !words_lt(load_words(pk, i, 32u * e), JJS_Q_WORDS) || !words_lt(load_words(rs, i, 32u * e), JJS_Q_WORDS) ||
                    !words_lt(load_words(ss, i, 32u * e), JJS_Q_WORDS);
    }
    words8 r2;
    for (int k = 0; k < 8; ++k) r2.w[k] = JJS_FR_R2_WORDS[k];
    const words8 cd = fr_mont_mul(fr_mont_mul(load_w8(P.c_words + 8 * t), r2), load_w8(P.d_words + 8 * i));
    build_point_table(ws, load_fq(pk, i), load_fq(pk, i, 32));
    ext_pt lhs = table_mul(ws, cd, true);                          // T needed by the comb additions
    lhs = add_comb(lhs, P.comb_g, z);
    const ext_pt e = load_ext(P.e_pt + EXT_WORDS * i);
    const bool ok = fq_eq(fq_mul(lhs.x, e.z), fq_mul(e.x, lhs.z)) && fq_eq(fq_mul(lhs.y, e.z), fq_mul(e.y, lhs.z));
    P.share_status[i] = (uint8_t)(malformed ? (uint32_t)ST_MALFORMED : (ok ? (uint32_t)ST_OK : (uint32_t)ST_INVALID_SHARE));
}

// pass 6 (lane per transcript): what `combine` returns (src/multisig.rs:326-360): the signature only when every
// share of the transcript verified; otherwise the status of the first failing share, and no signature -- the
// outputs are cleared so that a caller who ignores the statuses cannot pick up an aggregate built from bad shares.
// A transcript without participants is the reference's InvalidMultisigTranscript (src/multisig.rs:332-338): status 5 for
// that transcript alone, nothing out (its aggregate key is cleared as well).
JJS_HD void msig_verdict_item(const msig_params& P, uint32_t t) {
    uint32_t st = ST_OK;
    for (uint32_t i = P.offsets[t + 1]; i-- > P.offsets[t];) st = P.share_status[i] ? P.share_status[i] : st;
    if (P.offsets[t + 1] == P.offsets[t]) {
        st = ST_INVALID_TRANSCRIPT;
        store_words(P.agg_pk, 2 * (uint64_t)t, small_words(0));
        store_words(P.agg_pk, 2 * (uint64_t)t + 1, small_words(0));
    }
    if (P.transcript_status) P.transcript_status[t] = (uint8_t)st;
    if (st != ST_OK) {
        store_words(P.sig_u, t, small_words(0));
        store_words(P.sig_R, 2 * (uint64_t)t, small_words(0));
        store_words(P.sig_R, 2 * (uint64_t)t + 1, small_words(0));
    }
}

}  // namespace jjs
