// Width-5 Hades permutation + SAFE sponge: `Hash::digest_truncated(Domain::Other, inputs)[0]`.
//
// Replaces dusk-poseidon 0.42.0-rc.0 (+ dusk-safe) as called at /root/reference/src/signatures.rs:130,
// src/signatures/double.rs:162 and src/signatures/var_gen.rs:130.  Parameterisation per SURVEY.md A.3:
// 4 full + 60 partial + 4 full rounds, S-box x^5 on state[4] in partial rounds, round constants and
// Cauchy matrix as generated into jjs_constants.inc.  Constants are read at wave-uniform addresses
// (scalar cache, SGPR operands); each matrix row is ONE five-term dot product with a single
// Montgomery reduction (fq_dot_const) instead of five reduced products, and the 60 partial rounds use
// the controller-canonical form (two dot products per round).
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

// The permutation in its optimised form (constants and derivation: optimised_hades() in
// tools/gen_constants.py).  The 60 partial rounds are the time-invariant linear system
// p' = Mh p + v x, y' = w.p + m44 x driven by x = (y + kappa)^5; in controller canonical form a round is
// one S-box, TWO five-term dot products (new z3, new y) and a register shift, and only one constant is
// added.  The dense matrix appears in the 8 full rounds and in the last partial round, with the change
// of basis folded into the matrices on both sides of the partial block.  Same function of the state as
// the textbook round sequence (checked in the generator, on the host build and on the GPU).
JJS_HD void hades_permute(hades_state& st) {
    // nine dense-mix events: full rounds 0..3, the last partial round, full rounds 4..7
    for (int ev = 0; ev < 9; ++ev) {
        fe_n t[5];
        const uint32_t (*mat)[5][9];   // mat[i] = row i, a 5 x 9 block of limbs
        if (ev == 4) {
            for (int k = 0; k < 59; ++k) {
                fe_n v[5] = {st.s[0], st.s[1], st.s[2], st.s[3],
                             sbox5(fq_add(st.s[4], fe_from_const<1, 1>(JJS_HP_KAPPA[k])))};
                // new z3 then new y from ONE copy of the dot-product code (a rolled loop also keeps the
                // compiler from hoisting both constant rows into 90 SGPRs across the round loop)
                fe_n z3 = v[4], y = v[4];
#pragma unroll 1
                for (int r = 0; r < 2; ++r) {
                    z3 = y;
                    y = fq_dot_const<5, 2>(JJS_HP_ROWS[r], v);
                }
                st.s[0] = st.s[1]; st.s[1] = st.s[2]; st.s[2] = st.s[3]; st.s[3] = z3; st.s[4] = y;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = st.s[i];
            t[4] = sbox5(fq_add(st.s[4], fe_from_const<1, 1>(JJS_HP_KAPPA[59])));
            mat = JJS_HD_MAT[2];
        } else {
            const int fr = ev < 4 ? ev : ev - 1;
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = sbox5(fq_add(st.s[i], fe_from_const<1, 1>(JJS_HF_RC[fr][i])));
            mat = JJS_HD_MAT[ev == 3 ? 1 : 0];
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
