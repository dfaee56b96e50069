// Latency path for small batches (all three schemes): one signature spread over many lanes.
//
// The throughput path gives a signature one lane from start to finish, so a call costs one signature's
// latency (~600 k dependent instructions) however few items it carries.  What the reference's callers do --
// `PublicKey::verify` on one signature, a block's worth at a time (/root/reference/src/keys/public.rs:114) --
// is exactly that regime.  Here the same checks (same statuses, bit for bit) are cut so that the critical path is
//     challenge hash -> half-size scalars -> 1/4 of the double-and-add loop,
// and everything that does not depend on the challenge runs beside the hash on other lanes:
//
//   phase A, three roles in one launch (role = block range):
//     hash     8 lanes / item         encodings, Poseidon challenge (the five S-boxes of a full round on five
//                                     lanes), truncated Euclid (a, b)                                [prepare_item]
//     chain    8 lanes / equation     P_k = 2^(32k) * P for P in {PK, R}, k = 0..3, and the window table
//                                     {0..8} * P_k of each (the loop of phase B then needs no doubling chain
//                                     longer than 28); 16 lanes and 16-bit pieces for the smallest batches
//     point    1 lane / point         is_on_curve, !is_identity, is_torsion_free (pairing test) for every point:
//                                     no combined test, no resolve pass
//   phase B, 4 adjacent lanes / equation:
//     lane k   sum of the signed 4-bit windows 8k..8k+7 of a over table(PK_k) and of -b over table(R_k)
//              (28 doublings, 16 additions) plus comb digits 4k..4k+3 of (b*u)*G; the four partial sums are
//              added across the lanes (two shuffle rounds), the lanes of an item exchange their verdicts, and
//              lane 0 writes the status.
//   sum_k 2^(32k) * (a_k * PK - b_k * R) + (b*u) * G  is the left side of the half-size equation of verify_core.h
//   (check_equation), computed exactly for any curve points (complete addition law), so the statuses are the
//   reference's by the same argument; the only difference from the throughput path is that every point gets its
//   own subgroup test, which is the reference's `is_valid` literally (src/keys/public.rs:159-164).
//
// The per-item-generator scheme runs the same lanes on (c over PK, u over Gen) with full-size scalars (64 windows,
// chains up to 224 doublings: the chain, not the hash, is then the critical path) and compares the sum with R.
//
// ~300 k dependent instructions instead of ~600 k, for ~1.8 x the total work: used below a size threshold only.
#pragma once
#include "verify_core.h"
#include "ed29_quad.h"

namespace jjs {

// The 128-bit half-size scalars are cut into `positions` pieces (4, 8 or 16, chosen per launch: a finer cut halves
// the tail of the critical path again and doubles the work of the chain lanes, so it is used for the smaller
// batches only).  Piece k covers bits [128 k / positions, 128 (k + 1) / positions).
constexpr int SB_MAX_POSITIONS = 16;
static_assert(COMB_WINDOWS % SB_MAX_POSITIONS == 0, "the comb digits are shared out evenly");

struct small_params {
    verify_params V;        // scheme descriptor with small_mode = 1; V.prep holds the prep records
    uint32_t* tables;       // [item][equation][0 = PK, 1 = R][position][TABLE_WORDS]
    uint8_t* point_ok;      // [item][4]: V.points[p] is on the curve, not the identity, torsion-free
    uint32_t positions;     // 4, 8 or 16
    uint32_t windows;       // signed 4-bit windows of the scalars: 32 (half-size scalars, fixed generator) or 64 (per-item generator)
    uint32_t hash_lanes;    // 1, or SB_HASH_LANES: eight lanes share an item's challenge hash (smallest batches, where the
                            // hash is the critical path and the chip has lanes to spare)
    uint32_t quad_chains;   // the doublings of a chain lane on four lanes (sb_chain_lane_quad): small calls of the per-item-
                            // generator scheme, whose far positions (224 doublings) outlast the hash
};

JJS_HD uint32_t* sb_table(const small_params& S, uint64_t item, uint32_t e, uint32_t pt, uint32_t k) {
    return S.tables + ((((item * S.V.n_eq + e) * 2 + pt) * S.positions) + k) * (size_t)TABLE_WORDS;
}
JJS_HD size_t sb_table_words_per_item(uint32_t n_eq, uint32_t positions) { return (size_t)n_eq * 2 * positions * TABLE_WORDS; }

// table[j] = j * P for j = 0..8, P projective
JJS_HD void build_point_table_ext(uint32_t* tab, const ext_pt& p1) {
    const niels_pt n1 = to_niels(p1);
    store_niels(tab, niels_identity());
    store_niels(tab + ENTRY_WORDS, n1);
    ext_pt acc = ext_double(p1, true);
    store_niels(tab + 2 * ENTRY_WORDS, to_niels(acc));
    for (int j = 3; j <= 8; ++j) {
        acc = ext_add_niels(acc, n1, false, true);
        store_niels(tab + j * ENTRY_WORDS, to_niels(acc));
    }
}

// ---- phase A ------------------------------------------------------------------------------------------
constexpr int SB_HASH_LANES = 8;      // lanes that share one item's challenge hash on the device (hades_permute, coop)
JJS_HD void sb_hash_item(const small_params& S, uint64_t item) {
    store_prep(S.V.prep, S.V.n, item, prepare_item(S.V, item));
}
// the same on eight adjacent lanes (lane j of the group); every lane ends with the same record, lane 0 stores it
JJS_HD void sb_hash_item_coop(const small_params& S, uint64_t item, int j, bool active) {
    const prep_record r = prepare_item(S.V, item, j == 0 && active, j);
    if (j == 0 && active) store_prep(S.V.prep, S.V.n, item, r);
}
// window table of 2^(4 k windows / positions) * P; P = PK (pt 0), or (pt 1) R for a fixed-generator equation and the
// generator for the per-item-generator scheme
JJS_HD void sb_chain_lane(const small_params& S, uint64_t item, uint32_t e, uint32_t pt, uint32_t k) {
    const fe_src& src = pt == 0 ? S.V.eq[e].pk : (S.V.eq[e].comb ? S.V.eq[e].r : S.V.eq[e].gen);
    const fe_n pu = load_fq(src, item), pv = load_fq(src, item, 32);
    uint32_t* tab = sb_table(S, item, e, pt, k);
    if (k == 0) {                                    // wave-uniform: a launch gives every wave one position
        build_point_table(tab, pu, pv);
        return;
    }
    ext_pt p = ext_from_affine(pu, pv);
    const int n_dbl = 4 * ((int)S.windows / (int)S.positions) * (int)k;
    for (int i = 0; i < n_dbl; ++i) p = ext_double(p, i == n_dbl - 1);
    build_point_table_ext(tab, p);
}
#if defined(__HIPCC__)
// The same on a quad (lane j of four adjacent lanes): the doublings shared (ext_double_quad: a third of the chain's latency),
// the table built by lane 0.  Same products on the same limbs: the same table.
__device__ __forceinline__ void sb_chain_lane_quad(const small_params& S, uint64_t item, uint32_t e, uint32_t pt, uint32_t k, uint32_t j) {
    const fe_src& src = pt == 0 ? S.V.eq[e].pk : (S.V.eq[e].comb ? S.V.eq[e].r : S.V.eq[e].gen);
    const fe_n pu = load_fq(src, item), pv = load_fq(src, item, 32);
    uint32_t* tab = sb_table(S, item, e, pt, k);
    if (k == 0) {                                    // wave-uniform: a launch gives every wave one position
        if (j == 0) build_point_table(tab, pu, pv);
        return;
    }
    ext_pt p = ext_from_affine(pu, pv);
    const int n_dbl = 4 * ((int)S.windows / (int)S.positions) * (int)k;
    fe_n own;
#pragma unroll 1
    for (int i = 0; i < n_dbl; ++i) p = ext_double_quad(p, j, own);
    if (j == 0) build_point_table_ext(tab, p);
}
#endif
// `is_valid` of one point (src/keys/public.rs:159-164, src/signatures.rs:93-98)
JJS_HD void sb_point_lane(const small_params& S, uint64_t item, uint32_t p) {
    const fe_n pu = load_fq(S.V.points[p], item), pv = load_fq(S.V.points[p], item, 32);
    // points that come out of the wire decoder satisfy the curve equation by construction
    const bool ok = (S.V.decoded_points ? !affine_is_identity(pu, pv) : point_on_curve_not_identity(pu, pv)) &&
                    is_torsion_free(pu, pv);
    S.point_ok[4 * item + p] = ok ? 1 : 0;
}

// ---- phase B ------------------------------------------------------------------------------------------
// lane k of equation e: its windows / positions signed windows of both scalars over the tables at position k and, for a
// fixed generator, its 16 / positions comb digits of (b*u)*G.
//   fixed generator:    a over table(PK_k), -b over table(R_k)        (sum must be the identity)
//   per-item generator: c over table(PK_k),  u over table(Gen_k)      (sum must equal R)
JJS_HD ext_pt sb_piece(const small_params& S, uint64_t item, uint32_t e, uint32_t k, const prep_record& r) {
    const bool fixed = S.V.eq[e].comb != nullptr;
    const words8 u = load_words(S.V.u, item);
    const words8 s0 = fixed ? recode_signed4_128(r.h.a) : recode_signed4(r.c);
    const words8 s1 = fixed ? recode_signed4_128(r.h.b) : recode_signed4(u);
    const uint32_t* t0 = sb_table(S, item, e, 0, k);
    const uint32_t* t1 = sb_table(S, item, e, 1, k);
    const bool flip1 = fixed && !r.h.b_neg;                          // table 1 contributes -b*R
    const int windows = (int)S.windows / (int)S.positions, top = (int)S.windows - 1;
    ext_pt acc = ext_identity();
    for (int win = windows - 1; win >= 0; --win) {
        if (win != windows - 1) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) acc = ext_double(acc, j == 3);
        }
        const int nib = windows * (int)k + win;                      // digit `top` is the unsigned top digit
        acc = add_window(acc, t0, s0, nib, true, top, false);
        acc = add_window(acc, t1, s1, nib, !fixed || win == 0, top, flip1);   // T: for the comb digits / the lane sums
    }
    if (!fixed) return acc;
    const int comb_digits = COMB_WINDOWS / (int)S.positions;
    return add_comb_range(acc, S.V.eq[e].comb, half_scalar_times_u(r.h, u), comb_digits * (int)k, comb_digits * (int)k + comb_digits, true);
}
// the verdict of an equation from the sum of its pieces
JJS_HD bool sb_equation_holds(const small_params& S, uint64_t item, uint32_t e, const ext_pt& total) {
    if (S.V.eq[e].comb) return ext_is_identity(total);
    return ext_eq_affine(total, load_fq(S.V.eq[e].r, item), load_fq(S.V.eq[e].r, item, 32));
}
JJS_HD ext_pt sb_add(const ext_pt& a, const ext_pt& b) { return ext_add_niels(a, to_niels(b), false, true); }

JJS_HD uint32_t sb_status(bool malformed, bool points_ok, bool eq_ok) {
    return malformed ? ST_MALFORMED : (!points_ok ? ST_INVALID_POINT : (eq_ok ? ST_OK : ST_INVALID_SIGNATURE));
}
JJS_HD bool sb_points_ok(const small_params& S, uint64_t item) {
    bool ok = true;
    for (uint32_t p = 0; p < S.V.n_points; ++p) ok = ok && S.point_ok[4 * item + p] != 0;
    return ok;
}

// The whole path for one item on one thread: the CPU build's reference run of the same functions the device
// spreads over lanes (tests/hostbuild), with the partial sums added in the order of the device's shuffle tree.
JJS_HD uint32_t sb_verify_item_serial(const small_params& S, uint64_t item) {
    sb_hash_item(S, item);
    for (uint32_t e = 0; e < S.V.n_eq; ++e)
        for (uint32_t pt = 0; pt < 2; ++pt)
            for (uint32_t k = 0; k < S.positions; ++k) sb_chain_lane(S, item, e, pt, k);
    for (uint32_t p = 0; p < S.V.n_points; ++p) sb_point_lane(S, item, p);
    const prep_record r = load_prep(S.V.prep, S.V.n, item);
    bool eq_ok = true;
    for (uint32_t e = 0; e < S.V.n_eq; ++e) {
        ext_pt part[SB_MAX_POSITIONS];
        for (uint32_t k = 0; k < S.positions; ++k) part[k] = sb_piece(S, item, e, k, r);
        for (uint32_t step = 1; step < S.positions; step <<= 1)          // the device's shuffle tree: lane ^ 1, ^ 2, ^ 4
            for (uint32_t k = 0; k < S.positions; k += 2 * step) part[k] = sb_add(part[k], part[k + step]);
        const ext_pt total = part[0];
        eq_ok = sb_equation_holds(S, item, e, total) && eq_ok;
    }
    return sb_status(r.malformed, sb_points_ok(S, item), eq_ok);
}

}  // namespace jjs
