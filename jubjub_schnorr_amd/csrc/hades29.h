// Width-5 Hades permutation + SAFE sponge: `Hash::digest_truncated(Domain::Other, inputs)[0]`.
//
// Replaces dusk-poseidon 0.42.0-rc.0 (+ dusk-safe) as called at /root/reference/src/signatures.rs:130,
// src/signatures/double.rs:162 and src/signatures/var_gen.rs:130.  Parameterisation per SURVEY.md A.3:
// 4 full + 60 partial + 4 full rounds, S-box x^5 on state[4] in partial rounds, round constants and
// Cauchy matrix as generated into jjs_constants.inc.  Constants are read at wave-uniform addresses
// (scalar cache, SGPR operands); the matrix is applied in its small-integer form.
#pragma once
#include "fq29.h"

namespace jjs {

struct hades_state {
    fe_n s[5];
};

// x^5 for x = lane + round constant, taken as it comes out of the addition (limbs below 2^30): see
// fq_sqr_plus_const for why the first square needs no carry pass; the last product takes L = 2 anyway.
template <int A>
JJS_HD fe_n sbox5(const fe<2, A>& x) {
    fe_n x2 = fq_sqr_plus_const(x);
    fe_n x4 = fq_sqr(x2);
    return fq_mul(x4, x);
}

// The linear layer: st.s[i] = (sum_j S[i][j] * t[j]) / 2^29 with S[i][j] = JJS_HS_HANKEL[i + j].  On the
// device one asm block (tools/gen_mont_asm.py: one accumulator chain per row, the 9 distinct matrix entries
// as scalar operands); elsewhere five fq_lincomb_small calls.
JJS_HD void hades_matrix(hades_state& o, const fe_n (&t)[5]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t* h = JJS_HS_HANKEL;
    asm(JJS_HADES_MATRIX_ASM : JJS_HADES_MATRIX_OUTPUTS : JJS_HADES_MATRIX_INPUTS : JJS_HADES_MATRIX_CLOBBERS);
#else
#pragma unroll
    for (int i = 0; i < 5; ++i) o.s[i] = fq_lincomb_small<5>(JJS_HS_MAT[i], t);
#endif
}

// The permutation (constants and derivation: scaled_hades_constants() in tools/gen_constants.py).
// The MDS matrix is Cauchy, M[i][j] = F / (i + j + 5); with L = lcm(5..13) = 360360 it is (F/L) * S for a
// matrix S of INTEGERS below 2^17, so a matrix row is 45 small multiply-adds plus one Montgomery row
// (fq_lincomb_small) instead of a dot product with five 255-bit constants.  The scalar F/L * 2^29 is
// never multiplied in: the state is carried as s = lambda_r * s~ for a public per-round scale that passes
// through the S-box as lambda^5 and lives in the pre-scaled round constants; in a partial round lane 4
// is brought back to the common scale by one product with lambda_r^4, and constants on lanes 0..3 have
// been pushed forward so that only lane 4 receives one.  After round 68 the state is multiplied by
// lambda_end.  Same function of the state as the textbook round sequence (checked in the generator, on
// the host build and on the GPU).
//
// coop >= 0 (device only; the latency path, small_batch.h): eight adjacent lanes hold the same state and work on the
// same permutation.  In a full round lane j computes the S-box of state element min(j, 4) only and the five results are
// exchanged (45 ds_bpermute), so a full round costs one S-box instead of five on the critical path.  The linear layer is
// shared the same way in EVERY round: lane j computes row min(j, 4) of the matrix (fq_lincomb_small: 45 multiply-adds and
// one Montgomery row, 74 instructions instead of the 370 of all five rows) and the five rows are exchanged -- 60 partial
// rounds x ~250 instructions off the chain on which the whole call waits.  The S-box of a partial round (with the product
// that brings it back to the common scale) is x^5 mu = (x^2)^2 * (x mu): lanes 5..7 compute x mu while lanes 0..4 compute
// x^2, three products on the chain instead of four.  The products of this path are the inlined blocks (fq_mul_chain: no
// argument moves, no jump).  Same function of the same values: same digest.
// One call of 1 ... 1 024 single signatures: 0.49 -> 0.44-0.46 ms with the three-product S-box, -> 0.40-0.42 with the inlined
// blocks and the chains on quads (profiles/r04_small_call_chain.jsonl).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void hades_gather5(fe_n (&out)[5], const fe_n& mine, int base) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int w = 0; w < 9; ++w) out[i].l[w] = (uint32_t)__shfl((int)mine.l[w], base + i);
    }
}
#endif
JJS_HD void hades_permute(hades_state& st, int coop = -1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int cj = coop < 4 ? coop : 4, cbase = (int)(__lane_id() & ~7u);
    uint32_t crow[5] = {0, 0, 0, 0, 0};         // this lane's row of the matrix: S[cj][k] = JJS_HS_HANKEL[cj + k]
    if (coop >= 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) crow[k] = JJS_HS_HANKEL[cj + k];
    }
#endif
    for (int r = 0; r < 68; ++r) {
        fe_n t[5];
        if (r < 4 || r >= 64) {
            const int fr = r < 4 ? r : r - 60;
#if defined(__HIP_DEVICE_COMPILE__)
            if (coop >= 0) {
                fe_n x = st.s[4];
                fe_c rc = fe_from_const<1, 1>(JJS_HS_RC_FULL[fr][4]);
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    x = fq_select(cj == i, st.s[i], x);
                    rc = fq_select(cj == i, fe_from_const<1, 1>(JJS_HS_RC_FULL[fr][i]), rc);
                }
                const auto xc = fq_add(x, rc);
                hades_gather5(t, fq_mul_chain(fq_sqr_chain(fq_sqr_chain(xc)), xc), cbase);
                hades_gather5(st.s, fq_lincomb_small<5>(crow, t), cbase);
                continue;
            }
#endif
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = sbox5(fq_add(st.s[i], fe_from_const<1, 1>(JJS_HS_RC_FULL[fr][i])));
        } else {
            const int k = r - 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = st.s[i];
#if defined(__HIP_DEVICE_COMPILE__)
            if (coop >= 0) {
                // x^5 mu = (x^2)^2 * (x mu): three products on the chain instead of four.  Lanes 0..4 of the eight (the ones
                // whose matrix rows are read) take x^2 while lanes 5..7 take x mu WITH THE SAME INSTRUCTIONS (a product whose
                // second operand is chosen per lane); x mu travels to the others while they square again.  Lanes 5..7 end with
                // a t[4] that means nothing and a row nobody reads; the gather gives them the state back.
                const auto x = fq_add(st.s[4], fe_from_const<1, 1>(JJS_HS_KAPPA[k]));
                const fe_n p = fq_mul_chain(x, fq_select(coop >= 5, fq_as<2, 3>(fe_from_const<1, 1>(JJS_HS_MU[k])), x));
                fe_n y;
#pragma unroll
                for (int w = 0; w < 9; ++w) y.l[w] = (uint32_t)__shfl((int)p.l[w], cbase + 5);
                t[4] = fq_mul_chain(fq_sqr_chain(p), y);
                hades_gather5(st.s, fq_lincomb_small<5>(crow, t), cbase);
                continue;
            }
#endif
            // (out-of-line products here, where the chip is full: with the blocks inlined a 2^20 batch is 1.3 % slower -- partial
            // rounds only -- to 3.5 % -- every round: the code no longer fits beside the rest, profiles/r04_hash_inline_ab.txt)
            t[4] = fq_mul(sbox5(fq_add(st.s[4], fe_from_const<1, 1>(JJS_HS_KAPPA[k]))), fe_from_const<1, 1>(JJS_HS_MU[k]));
        }
        hades_matrix(st, t);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st.s[i] = fq_mul(st.s[i], fe_from_const<1, 1>(JJS_HS_LAMBDA_END));
}

// Absorbs n_inputs elements fetched through `fetch(e)` (which returns the Montgomery form of
// transcript element e) and squeezes one element.  The sponge is processed in rate-4 blocks so that
// the permutation has a single call site and no input has to stay live across a permutation.
template <typename Fetch>
JJS_HD fe_n poseidon_digest_tagged(int n_inputs, const fe_n& tag, Fetch fetch, int coop = -1) {
    hades_state st;
    st.s[0] = tag;
#pragma unroll
    for (int i = 1; i < 5; ++i) st.s[i] = fq_as<1, 2>(fq_zero());
    const int n_blocks = (n_inputs + 3) >> 2;
    for (int blk = 0; blk < n_blocks; ++blk) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int e = 4 * blk + k;
            if (e < n_inputs) st.s[1 + k] = fq_reduce(fq_norm(fq_add(st.s[1 + k], fetch(e))));
        }
        hades_permute(st, coop);
    }
    return st.s[1];
}

// transcripts of up to JJS_MAX_HASH_INPUTS elements: the SAFE tag comes from the constant table
template <typename Fetch>
JJS_HD fe_n poseidon_digest(int n_inputs, Fetch fetch, int coop = -1) {
    return poseidon_digest_tagged(n_inputs, fq_as<1, 2>(fe_from_const<1, 1>(JJS_SPONGE_TAG[n_inputs])), fetch, coop);
}

// digest -> canonical words, low 250 bits (JubJubScalar)
JJS_HD words8 truncate250(const fe_n& digest) {
    words8 c = fq_to_words(digest);
    c.w[7] &= 0x03ffffffu;
    return c;
}

}  // namespace jjs
