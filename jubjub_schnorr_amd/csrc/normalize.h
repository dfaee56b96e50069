// Extended-coordinate inputs: what `PublicKey::verify` actually receives is a `JubJubExtended`
// (/root/reference/src/keys/public.rs:114-118), and the reference normalises it itself with one field inversion
// per point (`to_hash_inputs()`, src/signatures.rs:127-128).  The *_ext entry points take the points as
// (U, V, Z) -- 96 bytes, three canonical field elements, u = U/Z, v = V/Z -- and do that normalisation on the
// device, so that a host shim never inverts anything.
//
// One inversion serves many items: a lane walks its items (item = lane, lane + lanes, ...), keeps the running
// product of their Z coordinates (one product per item covering its 2..4 points), stores each prefix in a scratch
// array, inverts the total once (~300 products) and unwinds backwards (Montgomery's trick): ~25 products per
// item plus 1/m of an inversion when a lane owns m items.
//
// Totality: a coordinate >= q marks the item malformed (status 3, like every other non-canonical encoding); Z = 0
// is not a point of the curve at all (the complete addition law never produces it): the item's affine output is
// (0, 0), which fails the curve equation in the verify kernel, i.e. InvalidPoint -- what `is_on_curve` says about
// such a value.  The T coordinates of the Rust type are not transferred: they are redundant for valid points.
#pragma once
#include "verify_core.h"

namespace jjs {

struct normalize_params {
    uint32_t n_src, pad_;
    fe_src src[4];        // extended points: U at off, V at off + 32, Z at off + 64 (stride 96 for plain arrays)
    uint8_t* out[4];      // affine u || v, n x 64 each
    uint8_t* bad;         // n bytes, set to 1 when a coordinate of item i is not canonical
    uint32_t* scratch;    // 9 words per item: running prefix products
    uint64_t n;           // items of this launch: first .. first + n - 1 (a host-buffer call normalises range by range)
    uint64_t first;
};

JJS_HD bool words_are_zero(const words8& a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) x |= a.w[i];
    return x == 0;
}

struct item_z {
    fe_n z[4];            // Z of each point in Montgomery form; 1 where Z = 0 or the slot is unused
    fe_n prod;            // their product
    uint32_t zero_mask;   // bit k: Z_k == 0
    bool malformed;       // some U, V or Z >= q
};

JJS_HD item_z load_item_z(const normalize_params& P, uint64_t item) {
    item_z r;
    r.zero_mask = 0;
    r.malformed = false;
    r.prod = fe_n_one();
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        r.z[k] = fe_n_one();
        if (k < P.n_src) {
            const words8 zw = load_words(P.src[k], item, 64);
            const bool zero = words_are_zero(zw), z_big = !words_lt(zw, JJS_Q_WORDS);
            r.malformed = r.malformed || z_big || !words_lt(load_words(P.src[k], item), JJS_Q_WORDS) ||
                          !words_lt(load_words(P.src[k], item, 32), JJS_Q_WORDS);
            r.zero_mask |= zero ? (1u << k) : 0u;
            // a Z that is zero or not canonical (q and 2q are zero mod q) must not enter the shared product
            r.z[k] = fq_select(zero || z_big, fe_n_one(), fq_from_words(zw));
            r.prod = (k == 0) ? r.z[0] : fq_mul(r.prod, r.z[k]);
        }
    }
    return r;
}

// All items of one lane: item = first + lane, first + lane + lanes, ...   (lanes = total number of lanes of the launch)
JJS_HD void normalize_lane(const normalize_params& P, uint64_t lane, uint64_t lanes) {
    if (lane >= P.n) return;
    const uint64_t count = (P.n - lane + lanes - 1) / lanes;
    fe_n acc = fe_n_one();
    for (uint64_t j = 0; j < count; ++j) {
        const uint64_t item = P.first + lane + j * lanes;
        uint32_t* s = P.scratch + 9 * item;
#pragma unroll
        for (int i = 0; i < 9; ++i) s[i] = acc.l[i];
        acc = fq_mul(acc, load_item_z(P, item).prod);
    }
    fe_n inv = fq_inverse(acc);                       // never zero: zero Z were replaced by 1
    for (uint64_t j = count; j-- > 0;) {
        const uint64_t item = P.first + lane + j * lanes;
        const item_z iz = load_item_z(P, item);
        fe_n pre;
        const uint32_t* s = P.scratch + 9 * item;
#pragma unroll
        for (int i = 0; i < 9; ++i) pre.l[i] = s[i];
        fe_n back = fq_mul(inv, pre);                 // 1 / (Z_0 ... Z_{k-1}) of this item
        inv = fq_mul(inv, iz.prod);
        // left[k] = Z_0 ... Z_{k-1}
        fe_n left[4];
        left[0] = fe_n_one();
#pragma unroll
        for (uint32_t k = 1; k < 4; ++k) left[k] = (k < P.n_src) ? (k == 1 ? iz.z[0] : fq_mul(left[k - 1], iz.z[k - 1])) : left[k - 1];
#pragma unroll
        for (uint32_t kk = 0; kk < 4; ++kk) {
            const uint32_t k = 3 - kk;
            if (k < P.n_src) {
                const fe_n zi = (k == 0) ? back : fq_mul(back, left[k]);          // 1 / Z_k
                if (k > 0) back = fq_mul(back, iz.z[k]);
                const bool zero = ((iz.zero_mask >> k) & 1u) != 0;
                words8 uw = fq_to_words(fq_mul(load_fq(P.src[k], item), zi));
                words8 vw = fq_to_words(fq_mul(load_fq(P.src[k], item, 32), zi));
#pragma unroll
                for (int i = 0; i < 8; ++i) { uw.w[i] = zero ? 0u : uw.w[i]; vw.w[i] = zero ? 0u : vw.w[i]; }
                store_words(P.out[k], 2 * item, uw);
                store_words(P.out[k], 2 * item + 1, vw);
            }
        }
        if (P.bad && iz.malformed) P.bad[item] = 1;
    }
}

}  // namespace jjs
