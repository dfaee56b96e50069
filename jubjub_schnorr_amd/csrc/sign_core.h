// Batch signing, one signature per lane: the generator of synthetic inputs for tests and bench.py.
// NOT constant time (table lookups and branches depend on the secret key): never use with
// production keys.  Results are bit-exact with the reference's signers given the same RNG draw:
//   SecretKey::sign          /root/reference/src/keys/secret.rs:174-194   nonce src/nonce.rs:32-44
//   SecretKey::sign_double   src/keys/secret/double.rs:56-85              nonce src/nonce.rs:49-61
//   SecretKeyVarGen::sign    src/keys/secret/var_gen.rs:228-256           nonce src/nonce.rs:68-85
#pragma once
#include "verify_core.h"

namespace jjs {

enum : uint32_t { SCHEME_SINGLE = 0, SCHEME_DOUBLE = 1, SCHEME_VARGEN = 2 };

struct sign_params {
    uint32_t scheme, pad_;
    const uint8_t *sk, *gen_scalar, *rnd, *m;
    uint8_t *u_out, *R_out, *Rp_out, *PK_out, *PKp_out, *Gen_out;
    const uint32_t *comb_g, *comb_gn;
    uint64_t n;
    uint32_t* workspace;  // WS_WORDS_PER_LANE words per resident lane
};

// ---- scalar multiplications --------------------------------------------------------------------
// k * Base from the fixed-base comb table (T of the result is not valid: callers only normalise it)
JJS_HD ext_pt comb_mul(const uint32_t* comb, const words8& k) { return add_comb(ext_identity(), comb, k); }
// k * P for the table of P built by build_point_table; k < 2^252.  T is only valid when final_t is set.
JJS_HD ext_pt table_mul(const uint32_t* tab, const words8& k, bool final_t = false) {
    const words8 sk = recode_signed4(k);
    ext_pt acc = ext_identity();
    for (int w = 63; w >= 0; --w) {
        if (w != 63) {
            acc = ext_double(acc, false);
            acc = ext_double(acc, false);
            acc = ext_double(acc, false);
            acc = ext_double(acc, true);
        }
        acc = add_window(acc, tab, sk, w, final_t && w == 0);
    }
    return acc;
}
struct affine_words {
    words8 u, v;
};
JJS_HD affine_words to_affine_words(const ext_pt& p) {
    fe_n zi = fq_inverse(p.z);
    affine_words a;
    a.u = fq_to_words(fq_mul(p.x, zi));
    a.v = fq_to_words(fq_mul(p.y, zi));
    return a;
}
JJS_HD void store_point(uint8_t* base, uint64_t item, const affine_words& a) {
    store_words(base, 2 * item, a.u);
    store_words(base, 2 * item + 1, a.v);
}

// transcript staging area: the lane's workspace tail, 16 x 32 bytes
JJS_HD void stage(uint32_t* area, int e, const words8& w) { store_words(reinterpret_cast<uint8_t*>(area), (uint64_t)e, w); }
JJS_HD words8 small_words(uint32_t x) {
    words8 w;
#pragma unroll
    for (int i = 0; i < 8; ++i) w.w[i] = 0;
    w.w[0] = x;
    return w;
}
JJS_HD words8 staged_digest(const uint32_t* area, int n) {
    fe_src s{reinterpret_cast<const uint8_t*>(area), 0, 0};
    fe_n d = poseidon_digest(n, [&](int e) { return load_fq(s, 0, 32u * (uint32_t)e); });
    return truncate250(d);
}

JJS_HD void sign_item(const sign_params& P, uint64_t item, uint32_t* ws) {
    uint32_t* tab = ws;                    // one point table
    uint32_t* area = ws + TABLE_WORDS;     // transcript staging (second table's space)
    const fe_src s_sk{P.sk, 32, 0}, s_rnd{P.rnd, 32, 0}, s_m{P.m, 32, 0};
    const words8 sk = load_words(s_sk, item), rnd = load_words(s_rnd, item), m = load_words(s_m, item);

    if (P.scheme == SCHEME_VARGEN) {
        const fe_src s_g{P.gen_scalar, 32, 0};
        affine_words gen = to_affine_words(comb_mul(P.comb_g, load_words(s_g, item)));
        stage(area, 0, rnd); stage(area, 1, sk); stage(area, 2, gen.u); stage(area, 3, gen.v); stage(area, 4, m);
        const words8 r = staged_digest(area, 5);
        build_point_table(tab, fq_from_words(gen.u), fq_from_words(gen.v));
        affine_words R = to_affine_words(table_mul(tab, r));
        affine_words PK = to_affine_words(table_mul(tab, sk));
        stage(area, 0, R.u); stage(area, 1, R.v); stage(area, 2, PK.u); stage(area, 3, PK.v);
        stage(area, 4, gen.u); stage(area, 5, gen.v); stage(area, 6, m);
        const words8 c = staged_digest(area, 7);
        store_words(P.u_out, item, fr_sub_mul(r, c, sk));
        store_point(P.R_out, item, R); store_point(P.PK_out, item, PK); store_point(P.Gen_out, item, gen);
        return;
    }
    const bool dbl = (P.scheme == SCHEME_DOUBLE);
    stage(area, 0, rnd); stage(area, 1, sk); stage(area, 2, small_words(dbl ? 2u : 1u)); stage(area, 3, m);
    const words8 r = staged_digest(area, 4);
    affine_words R = to_affine_words(comb_mul(P.comb_g, r));
    affine_words PK = to_affine_words(comb_mul(P.comb_g, sk));
    words8 c;
    if (dbl) {
        affine_words Rp = to_affine_words(comb_mul(P.comb_gn, r));
        affine_words PKp = to_affine_words(comb_mul(P.comb_gn, sk));
        words8 tag;
#pragma unroll
        for (int i = 0; i < 8; ++i) tag.w[i] = JJS_DOUBLE_TAG_WORDS[i];
        stage(area, 0, tag);
        stage(area, 1, R.u); stage(area, 2, R.v); stage(area, 3, Rp.u); stage(area, 4, Rp.v);
        stage(area, 5, PK.u); stage(area, 6, PK.v); stage(area, 7, PKp.u); stage(area, 8, PKp.v); stage(area, 9, m);
        c = staged_digest(area, 10);
        store_point(P.Rp_out, item, Rp); store_point(P.PKp_out, item, PKp);
    } else {
        stage(area, 0, R.u); stage(area, 1, R.v); stage(area, 2, PK.u); stage(area, 3, PK.v); stage(area, 4, m);
        c = staged_digest(area, 5);
    }
    store_words(P.u_out, item, fr_sub_mul(r, c, sk));
    store_point(P.R_out, item, R); store_point(P.PK_out, item, PK);
}

}  // namespace jjs
