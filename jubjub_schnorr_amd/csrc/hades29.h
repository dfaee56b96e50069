// Width-5 Hades permutation + SAFE sponge: `Hash::digest_truncated(Domain::Other, inputs)[0]`.
//
// Replaces dusk-poseidon 0.42.0-rc.0 (+ dusk-safe) as called at /root/reference/src/signatures.rs:130,
// src/signatures/double.rs:162 and src/signatures/var_gen.rs:130.  Parameterisation per SURVEY.md A.3:
// 4 full + 60 partial + 4 full rounds, S-box x^5 on state[4] in partial rounds, round constants and
// Cauchy matrix as generated into jjs_constants.inc.  Constants are read at wave-uniform addresses
// (scalar cache, SGPR operands); each matrix row is ONE five-term dot product with a single
// Montgomery reduction (fq_dot_const) instead of five reduced products, and the 60 partial rounds use
// the sparse-matrix form.
#pragma once
#include "fq29.h"

namespace jjs {

struct hades_state {
    fe_n s[5];
};

template <int L, int A>
JJS_HD fe_n sbox5(const fe<L, A>& x) {
    fe_n x2 = fq_sqr(x);
    fe_n x4 = fq_sqr(x2);
    return fq_mul(x4, x);
}

// The permutation in its optimised form (constants: optimised_hades() in tools/gen_constants.py):
// every partial round adds ONE constant (lane 4), applies the S-box to lane 4 and a sparse matrix --
// lane 4 becomes a five-term dot product, lanes 0..3 each gain col[i] * sbox -- and the dense matrix
// appears only in the 8 full rounds and after the last partial round.  Same function of the state as
// the textbook round sequence (checked in the generator, on the host build and on the GPU).
JJS_HD void hades_permute(hades_state& st) {
    // nine dense-mix events: full rounds 0..3, the last partial round, full rounds 4..7
    for (int ev = 0; ev < 9; ++ev) {
        fe_n t[5];
        const uint32_t (*mat)[5][9];   // mat[i] = row i, a 5 x 9 block of limbs
        if (ev == 4) {
            for (int k = 0; k < 59; ++k) {
                fe_n x4 = sbox5(fq_add(st.s[4], fe_from_const<1, 1>(JJS_HP_KAPPA[k])));
                fe_n v[5] = {st.s[0], st.s[1], st.s[2], st.s[3], x4};
                fe_n n4 = fq_dot_const<5, 2>(JJS_HP_ROW[k], v);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    st.s[i] = fq_reduce(fq_norm(fq_add(st.s[i], fq_mul(x4, fe_from_const<1, 1>(JJS_HP_COL[k][i])))));
                st.s[4] = n4;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = st.s[i];
            t[4] = sbox5(fq_add(st.s[4], fe_from_const<1, 1>(JJS_HP_KAPPA[59])));
            mat = JJS_HP_LAST;
        } else {
            const int fr = ev < 4 ? ev : ev - 1;
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = sbox5(fq_add(st.s[i], fe_from_const<1, 1>(JJS_HF_RC[fr][i])));
            mat = JJS_MDS;
        }
        // One copy of the dot-product code: rows are produced in order into s[4] while the state
        // registers rotate down, so after five steps s[i] holds row i (static indices only).
#pragma unroll 1
        for (int i = 0; i < 5; ++i) {
            fe_n row = fq_dot_const<5, 2>(mat[i], t);
            st.s[0] = st.s[1]; st.s[1] = st.s[2]; st.s[2] = st.s[3]; st.s[3] = st.s[4]; st.s[4] = row;
        }
    }
}

// Absorbs n_inputs elements fetched through `fetch(e)` (which returns the Montgomery form of
// transcript element e) and squeezes one element.  The sponge is processed in rate-4 blocks so that
// the permutation has a single call site and no input has to stay live across a permutation.
template <typename Fetch>
JJS_HD fe_n poseidon_digest(int n_inputs, Fetch fetch) {
    hades_state st;
    st.s[0] = fq_as<1, 2>(fe_from_const<1, 1>(JJS_SPONGE_TAG[n_inputs]));
#pragma unroll
    for (int i = 1; i < 5; ++i) st.s[i] = fq_as<1, 2>(fq_zero());
    const int n_blocks = (n_inputs + 3) >> 2;
    for (int blk = 0; blk < n_blocks; ++blk) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int e = 4 * blk + k;
            if (e < n_inputs) st.s[1 + k] = fq_reduce(fq_norm(fq_add(st.s[1 + k], fetch(e))));
        }
        hades_permute(st);
    }
    return st.s[1];
}

// digest -> canonical words, low 250 bits (JubJubScalar)
JJS_HD words8 truncate250(const fe_n& digest) {
    words8 c = fq_to_words(digest);
    c.w[7] &= 0x03ffffffu;
    return c;
}

}  // namespace jjs
