// CPU build of the SAME arithmetic / verification source the HIP kernels compile
// (jubjub_schnorr_amd/csrc/*.h), for `-m "not gpu"` tests and sanitizer runs.  Test
// infrastructure: the product library (libjjs_gpu.so) never contains or calls this.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "schemes.h"
#include "decode.h"
#include "normalize.h"
#include "small_batch.h"
#include "key_tables.h"
#include <map>
#include <string>
#include "multisig_core.h"
#include "jjs_sponge_tags_long.inc"
#include "safe_tag.h"

using namespace jjs;

static std::vector<uint32_t> g_comb_g, g_comb_gn;
alignas(16) static uint32_t g_tag[8];

// Comb tables for the CPU build.  The device builds every entry from scratch in its own lane
// (build_comb_entry); one CPU thread fills a row by repeated addition of 2^(COMB_BITS*i)*Base and one
// batched inversion instead -- the same values (tests compare sampled entries of both with the oracle,
// and a handful of entries here with build_comb_entry itself).
static void fill_comb_row(uint32_t* table, const uint32_t (*base)[9], int i) {
    fe_n bu = fq_as<1, 2>(fe_from_const<1, 1>(base[0])), bv = fq_as<1, 2>(fe_from_const<1, 1>(base[1]));
    ext_pt step = ext_from_affine(bu, bv);
    for (int k = 0; k < COMB_BITS * i; ++k) step = ext_double(step, true);
    const niels_pt nstep = to_niels(step);
    std::vector<ext_pt> pts(COMB_ENTRIES);
    std::vector<fe_n> prefix(COMB_ENTRIES);
    ext_pt acc = ext_identity();
    fe_n run = fe_n_one();
    for (int b = 0; b < COMB_ENTRIES; ++b) {
        pts[b] = acc;
        prefix[b] = run;                       // product of Z_0 .. Z_{b-1}
        run = fq_mul(run, acc.z);
        acc = ext_add_niels(acc, nstep, false, true);
    }
    fe_n inv = fq_inverse(run);                // 1 / (Z_0 ... Z_{last})
    for (int b = COMB_ENTRIES - 1; b >= 0; --b) {
        fe_n zi = fq_mul(inv, prefix[b]);
        inv = fq_mul(inv, pts[b].z);
        fe_n x = fq_mul(pts[b].x, zi), y = fq_mul(pts[b].y, zi);
        fe_n ypx = fq_reduce(fq_norm(fq_add(y, x)));
        fe_n ymx = fq_mul(fq_norm(fq_sub(y, x)), fq_one());
        fe_n t2d = fq_mul(fq_mul(x, y), fe_from_const<1, 1>(JJS_D2));
        uint32_t* dst = table + ((size_t)i * COMB_ENTRIES + b) * COMB_ENTRY_WORDS;
        for (int k = 0; k < 9; ++k) { dst[k] = ypx.l[k]; dst[9 + k] = ymx.l[k]; dst[18 + k] = t2d.l[k]; }
        dst[27] = 0;
    }
}

static void ensure_tables() {
    if (!g_comb_g.empty()) return;
    g_comb_g.resize(COMB_TABLE_WORDS);
    g_comb_gn.resize(COMB_TABLE_WORDS);
    for (int i = 0; i < COMB_WINDOWS; ++i) {
        fill_comb_row(g_comb_g.data(), JJS_G, i);
        fill_comb_row(g_comb_gn.data(), JJS_GN, i);
    }
    memcpy(g_tag, JJS_DOUBLE_TAG_WORDS, 32);
}

// jjs_host_set_split_prepare(1): the first pass in the two launches the device uses while the keys of a batch are still
// being counted (PREP_HEAD, the record through its buffer, PREP_TAIL), instead of PREP_ALL
static int g_split_prepare = 0;
static void run(verify_params P) {
    std::vector<uint32_t> ws(WS_WORDS_PER_LANE + 4);
    uint32_t* w = (uint32_t*)(((uintptr_t)ws.data() + 15) & ~(uintptr_t)15);
    std::vector<uint8_t> prep(65 * P.n + 64);
    uint32_t not_keyed = 0;
    if (g_split_prepare) {
        P.key_flag = &not_keyed;
        for (uint64_t i = 0; i < P.n; ++i) store_prep(prep.data(), P.n, i, prepare_item(P, i, true, -1, PREP_HEAD));
        for (uint64_t i = 0; i < P.n; ++i) store_prep(prep.data(), P.n, i, prepare_tail(P, i, load_prep(prep.data(), P.n, i)));
    }
    for (uint64_t i = 0; i < P.n; ++i) {
        uint32_t st = g_split_prepare ? finish_item(P, i, w, load_prep(prep.data(), P.n, i)) : verify_item(P, i, w);
        if (st >= ST_PENDING_EQ_FAILED) st = resolve_item(P, i, st == ST_PENDING_EQ_HELD);
        if (P.status) P.status[i] = (uint8_t)st;
        if (P.tally) P.tally[st]++;
    }
}

// the latency path of csrc/small_batch.h, every role of an item run in turn on this thread
static void run_small(verify_params P, uint32_t positions) {
    std::vector<uint8_t> prep(65 * P.n + 64), ok(4 * P.n + 4);
    std::vector<uint32_t> tables(sb_table_words_per_item(P.n_eq, positions) * P.n + 4);
    small_params S{};
    S.positions = positions;
    S.windows = P.eq[0].comb ? 32 : 64;
    P.small_mode = 1;
    P.prep = (uint8_t*)(((uintptr_t)prep.data() + 15) & ~(uintptr_t)15);
    S.V = P;
    S.tables = (uint32_t*)(((uintptr_t)tables.data() + 15) & ~(uintptr_t)15);
    S.point_ok = ok.data();
    for (uint64_t i = 0; i < P.n; ++i) {
        const uint32_t st = sb_verify_item_serial(S, i);
        if (P.status) P.status[i] = (uint8_t)st;
        if (P.tally) P.tally[st]++;
    }
}

// the key-table path of csrc/key_tables.h: keys deduplicated with a std::map (the device uses a hash table with
// the same byte-exact notion of "same key"), everything else through the product's own functions
static int g_key_window = KT_WINDOW_NARROW;      // jjs_host_set_key_window: both widths of the product run here
static void run_keyed(verify_params P) {
    const int w = g_key_window;
    key_params K{};
    K.n = P.n; K.max_keys = (uint32_t)P.n + 1;
    fe_src cols[2]; uint32_t n_cols = 0;
    for (uint32_t e = 0; e < P.n_eq; ++e) {
        cols[P.eq[e].pk_col] = P.eq[e].pk; n_cols = std::max(n_cols, (uint32_t)P.eq[e].pk_col + 1);
        if (!P.eq[e].comb) { cols[P.eq[e].gen_col] = P.eq[e].gen; n_cols = std::max(n_cols, (uint32_t)P.eq[e].gen_col + 1); }
    }
    K.n_cols = n_cols;
    std::vector<uint32_t> counters(64, 0), keyid[2], key_item[2], bases[2], tables[2];
    std::vector<uint8_t> flags[2];
    K.counters = counters.data();
    for (uint32_t c = 0; c < n_cols; ++c) {
        key_column& C = K.col[c];
        C.src = cols[c];
        keyid[c].resize(P.n + 1);
        std::map<std::string, uint32_t> seen;
        for (uint64_t i = 0; i < P.n; ++i) {
            std::string key((const char*)(C.src.base + i * C.src.stride + C.src.off), 64);
            auto it = seen.find(key);
            if (it == seen.end()) { it = seen.emplace(key, (uint32_t)seen.size()).first; key_item[c].push_back((uint32_t)i); }
            keyid[c][i] = it->second;
        }
        const size_t nk = key_item[c].size();
        counters[c] = (uint32_t)nk;
        // (the pool of the device is not cleared between calls: an invalid key's tables, which nobody builds, are whatever lay there)
        flags[c].resize(nk + 1); bases[c].resize(nk * kt_positions(w) * KT_BASE_WORDS + 4); tables[c].assign(nk * kt_positions(w) * kt_table_words(w) + 8, 0xA5C3F00Du);
        C.keyid = keyid[c].data(); C.key_item = key_item[c].data(); C.key_flags = flags[c].data();
        C.bases = bases[c].data();
        C.tables = (uint32_t*)(((uintptr_t)tables[c].data() + 15) & ~(uintptr_t)15);
        for (uint32_t id = 0; id < nk; ++id) {
            if (!kt_chain_key(C, id, w)) continue;           // as key_chain_kernel / key_table_kernel: tables for the valid keys only
            for (uint32_t pos = 0; pos < (uint32_t)kt_positions(w); ++pos) kt_table_lane(C, id, pos, w);
        }
    }
    counters[2] = (uint32_t)w;
    P.key_flag = &counters[2];
    for (uint64_t i = 0; i < P.n; ++i) {
        // the head launch alone is what the key-table path gets on the device (the tail leaves at once)
        uint32_t st = kt_finish_item(P, K, i, g_split_prepare ? prepare_item(P, i, true, -1, PREP_HEAD) : prepare_item(P, i));
        if (st >= ST_PENDING_EQ_FAILED) st = resolve_item(P, i, st == ST_PENDING_EQ_HELD);
        if (P.status) P.status[i] = (uint8_t)st;
        if (P.tally) P.tally[st]++;
    }
}

extern "C" {

int jjs_host_set_split_prepare(int on) {
    g_split_prepare = on ? 1 : 0;
    return 0;
}
int jjs_host_set_key_window(int w) {
    if (w != KT_WINDOW_NARROW && w != KT_WINDOW_WIDE) return -1;
    g_key_window = w;
    return 0;
}
int jjs_host_verify_keyed_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n,
                                 uint8_t* status, uint64_t* tally) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run_keyed(params_single(u, R, PK, m, n, g_comb_g.data(), out_ptrs{status, t, nullptr, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
int jjs_host_verify_keyed_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                                 const uint8_t* m, size_t n, uint8_t* status, uint64_t* tally) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run_keyed(params_double(u, R, Rp, PK, PKp, m, n, (const uint8_t*)g_tag, g_comb_g.data(), g_comb_gn.data(),
                            out_ptrs{status, t, nullptr, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
int jjs_host_verify_keyed_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                                 size_t n, uint8_t* status, uint64_t* tally) {
    unsigned long long t[4] = {0, 0, 0, 0};
    run_keyed(params_vargen(u, R, PK, Gen, m, n, out_ptrs{status, t, nullptr, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}

int jjs_host_verify_small_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n,
                                 uint8_t* status, uint64_t* tally, int positions) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run_small(params_single(u, R, PK, m, n, g_comb_g.data(), out_ptrs{status, t, nullptr, nullptr}), (uint32_t)positions);
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
int jjs_host_verify_small_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                                 const uint8_t* m, size_t n, uint8_t* status, uint64_t* tally, int positions) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run_small(params_double(u, R, Rp, PK, PKp, m, n, (const uint8_t*)g_tag, g_comb_g.data(), g_comb_gn.data(),
                            out_ptrs{status, t, nullptr, nullptr}), (uint32_t)positions);
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}

int jjs_host_verify_small_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                                 size_t n, uint8_t* status, uint64_t* tally, int positions) {
    unsigned long long t[4] = {0, 0, 0, 0};
    run_small(params_vargen(u, R, PK, Gen, m, n, out_ptrs{status, t, nullptr, nullptr}), (uint32_t)positions);
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}

int jjs_host_fq_mul(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 x, y;
        memcpy(x.w, a + 32 * i, 32); memcpy(y.w, b + 32 * i, 32);
        fe_n r = fq_mul(fq_from_words(x), fq_from_words(y));
        words8 o = fq_to_words(r);
        memcpy(out + 32 * i, o.w, 32);
    }
    return 0;
}
int jjs_host_fq_sqr(const uint8_t* a, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 x;
        memcpy(x.w, a + 32 * i, 32);
        words8 o = fq_to_words(fq_sqr(fq_from_words(x)));
        memcpy(out + 32 * i, o.w, 32);
    }
    return 0;
}
// out = a - b, a + b (64 bytes per item), through the lazy add/sub/norm/reduce path
int jjs_host_fq_addsub(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 x, y;
        memcpy(x.w, a + 32 * i, 32); memcpy(y.w, b + 32 * i, 32);
        fe_n fx = fq_from_words(x), fy = fq_from_words(y);
        words8 d = fq_to_words(fq_norm(fq_sub(fx, fy)));
        words8 s = fq_to_words(fq_reduce(fq_norm(fq_add(fx, fy))));
        memcpy(out + 64 * i, d.w, 32); memcpy(out + 64 * i + 32, s.w, 32);
    }
    return 0;
}
int jjs_host_fq_inv(const uint8_t* a, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 x;
        memcpy(x.w, a + 32 * i, 32);
        words8 o = fq_to_words(fq_inverse(fq_from_words(x)));
        memcpy(out + 32 * i, o.w, 32);
    }
    return 0;
}
int jjs_host_poseidon(const uint8_t* in, size_t k, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        fe_n d = poseidon_digest((int)k, [&](int e) {
            words8 x;
            memcpy(x.w, in + 32 * (k * i + e), 32);
            return fq_from_words(x);
        });
        words8 o = fq_to_words(d);
        memcpy(out + 32 * i, o.w, 32);
    }
    return 0;
}
// bit0 on_curve, bit1 torsion_free (pairing test), bit2 identity, bit3 torsion_free ([r]P == O)
int jjs_host_point_flags(const uint8_t* P, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 x, y;
        memcpy(x.w, P + 64 * i, 32); memcpy(y.w, P + 64 * i + 32, 32);
        fe_n u = fq_from_words(x), v = fq_from_words(y);
        bool id = affine_is_identity(u, v);
    out[i] = (uint8_t)((affine_on_curve(u, v) ? 1 : 0) | ((id || is_torsion_free(u, v)) ? 2 : 0) | (id ? 4 : 0) |
                       (is_torsion_free_by_order(u, v) ? 8 : 0));
    }
    return 0;
}
// c (n x 32) -> a (16 bytes) || |b| (16 bytes) || sign byte, 33 bytes per item
int jjs_host_half_size(const uint8_t* c, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        words8 w;
        memcpy(w.w, c + 32 * i, 32);
        half_scalars h = half_size_scalars(w);
        memcpy(out + 33 * i, h.a.w, 16); memcpy(out + 33 * i + 16, h.b.w, 16);
        out[33 * i + 32] = h.b_neg ? 1 : 0;
    }
    return 0;
}
// compressed (n x 32) -> affine (n x 64) + ok byte
int jjs_host_decompress(const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
    static std::vector<uint32_t> pw;
    static std::vector<uint8_t> hs;
    if (pw.empty()) {
        pw.resize(DLOG_POW_WORDS); hs.assign(65536, 0);
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 256; ++j) dlog_table_entry(pw.data(), hs.data(), i, j);
    }
    const dlog_tables T{pw.data(), hs.data()};
    for (size_t i = 0; i < n; ++i) {
        words8 w;
        memcpy(w.w, in + 32 * i, 32);
        decoded_point d = decompress_point(w, T);
        memcpy(out + 64 * i, d.u.w, 32); memcpy(out + 64 * i + 32, d.v.w, 32);
        ok[i] = d.ok ? 1 : 0;
    }
    return 0;
}
// extended points (k arrays of n x 96) -> affine (k arrays of n x 64) + malformed flags, run as `lanes` lanes
int jjs_host_normalize(const uint8_t* const* ext, int k, size_t n, size_t lanes, uint8_t* const* out, uint8_t* bad) {
    std::vector<uint32_t> scratch(9 * n + 16);
    normalize_params P{};
    P.n_src = (uint32_t)k; P.n = n; P.bad = bad; P.scratch = scratch.data();
    for (int i = 0; i < k; ++i) { P.src[i] = fe_src{ext[i], 96, 0}; P.out[i] = out[i]; }
    memset(bad, 0, n);
    for (size_t lane = 0; lane < lanes; ++lane) normalize_lane(P, lane, lanes);
    return 0;
}
int jjs_host_multisig(const uint8_t* z, const uint8_t* PK, const uint8_t* R, const uint8_t* S, const uint8_t* m,
                      const uint32_t* offsets, size_t B, uint8_t* status, uint8_t* agg_pk, uint8_t* sig_u, uint8_t* sig_R,
                      uint8_t* transcript_status) {
    ensure_tables();
    const size_t n = offsets[B];
    std::vector<uint32_t> tr(n), d(8 * n), dpk(EXT_WORDS * n), ept(EXT_WORDS * n), a(8 * B), c(8 * B), ws(WS_WORDS_PER_LANE + 4);
    msig_params P{};
    P.z = z; P.PK = PK; P.R = R; P.S = S; P.m = m; P.offsets = offsets; P.n_transcripts = (uint32_t)B; P.n_total = n;
    P.share_status = status; P.agg_pk = agg_pk; P.sig_u = sig_u; P.sig_R = sig_R; P.transcript_status = transcript_status;
    P.tr_of = tr.data(); P.d_words = d.data(); P.dpk = dpk.data(); P.e_pt = ept.data(); P.a_words = a.data(); P.c_words = c.data();
    P.tags = &JJS_SPONGE_TAG_LONG[0][0]; P.comb_g = g_comb_g.data();
    // transcripts beyond the tag table: pass 0 computes the two tags per transcript (csrc/safe_tag.h), as on the device
    std::vector<uint32_t> long_tags(18 * B + 18, 0u);
    P.max_table_participants = JJS_MSIG_MAX_PARTICIPANTS;
    P.long_tags = long_tags.data();
    uint32_t* w = (uint32_t*)(((uintptr_t)ws.data() + 15) & ~(uintptr_t)15);
    for (size_t t = 0; t < B; ++t) msig_map_item(P, (uint32_t)t);
    for (size_t i = 0; i < n; ++i) msig_delin_item(P, i, w);
    for (size_t t = 0; t < B; ++t) msig_agg_item(P, (uint32_t)t);
    for (size_t i = 0; i < n; ++i) msig_commit_item(P, i, w);
    for (size_t t = 0; t < B; ++t) msig_final_item(P, (uint32_t)t);
    for (size_t i = 0; i < n; ++i) msig_share_item(P, i, w);
    for (size_t t = 0; t < B; ++t) msig_verdict_item(P, (uint32_t)t);
    return 0;
}
// csrc/safe_tag.h: the SAFE tag of an n-element transcript as nine Montgomery limbs, and the generated table's row
int jjs_host_safe_tag(uint32_t n_inputs, uint32_t* out9) {
    safe_tag_limbs(n_inputs, JJS_Q_WORDS, out9);
    return 0;
}
int jjs_host_safe_tag_table(uint32_t n_inputs, uint32_t* out9) {
    if (n_inputs >= JJS_LONG_TAGS) return -1;
    memcpy(out9, JJS_SPONGE_TAG_LONG[n_inputs], 36);
    return 0;
}
// raw entry points on arbitrary (un-normalised) limb vectors, for the bound-edge tests: n x 9 uint32 each
int jjs_host_raw_mul(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
    for (size_t i = 0; i < n; ++i) {
        raw9 r = mont_mul_body(a + 9 * i, b + 9 * i);
        memcpy(out + 9 * i, r.l, 36);
    }
    return 0;
}
int jjs_host_raw_sqr(const uint32_t* a, size_t n, uint32_t* out) {
    for (size_t i = 0; i < n; ++i) {
        raw9 r = mont_sqr_body(a + 9 * i);
        memcpy(out + 9 * i, r.l, 36);
    }
    return 0;
}
// five-term dot product with the MDS row `row` and small-integer combination with JJS_HS_MAT[row]
int jjs_host_raw_dot5(const uint32_t* t, int row, size_t n, uint32_t* out_dot, uint32_t* out_small) {
    for (size_t i = 0; i < n; ++i) {
        fe<1, 3> v[5];
        for (int j = 0; j < 5; ++j) memcpy(v[j].l, t + 45 * i + 9 * j, 36);
        fe_n d = fq_dot_const<5, 3>(JJS_MDS[row], v);
        memcpy(out_dot + 9 * i, d.l, 36);
        fe<2, 3> w[5];
        for (int j = 0; j < 5; ++j) memcpy(w[j].l, t + 45 * i + 9 * j, 36);
        fe_n s = fq_lincomb_small<5>(JJS_HS_MAT[row], w);
        memcpy(out_small + 9 * i, s.l, 36);
    }
    return 0;
}
int jjs_host_comb_bits(void) { return COMB_BITS; }
// the device's per-entry builder on one entry: 1 if it equals the row-filled table entry
int jjs_host_comb_entry_matches_device_builder(int which, int i, int b) {
    ensure_tables();
    uint32_t scratch[COMB_ENTRY_WORDS];
    comb_entry_words(scratch, which ? JJS_GN : JJS_G, i, b);
    const uint32_t* t = (which ? g_comb_gn.data() : g_comb_g.data()) + ((size_t)i * COMB_ENTRIES + b) * COMB_ENTRY_WORDS;
    for (int c = 0; c < 3; ++c) {
        fe_n f1, f2;
        for (int k = 0; k < 9; ++k) { f1.l[k] = t[9 * c + k]; f2.l[k] = scratch[9 * c + k]; }
        if (!fq_eq(f1, f2)) return 0;
    }
    return 1;
}
// comb table entry -> affine point bytes (u || v), recovered from the cached form
int jjs_host_comb_entry(int which, int i, int b, uint8_t* out_ypx_ymx_t2d) {
    ensure_tables();
    const uint32_t* t = (which ? g_comb_gn.data() : g_comb_g.data()) + ((size_t)i * COMB_ENTRIES + b) * COMB_ENTRY_WORDS;
    for (int c = 0; c < 3; ++c) {
        fe_n f;
        for (int k = 0; k < 9; ++k) f.l[k] = t[9 * c + k];
        words8 o = fq_to_words(f);
        memcpy(out_ypx_ymx_t2d + 32 * c, o.w, 32);
    }
    return 0;
}
int jjs_host_verify_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n,
                           uint8_t* status, uint64_t* tally, uint8_t* c_out) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run(params_single(u, R, PK, m, n, g_comb_g.data(), out_ptrs{status, t, c_out, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
int jjs_host_verify_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                           const uint8_t* m, size_t n, uint8_t* status, uint64_t* tally, uint8_t* c_out) {
    ensure_tables();
    unsigned long long t[4] = {0, 0, 0, 0};
    run(params_double(u, R, Rp, PK, PKp, m, n, (const uint8_t*)g_tag, g_comb_g.data(), g_comb_gn.data(),
                      out_ptrs{status, t, c_out, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
int jjs_host_verify_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                           size_t n, uint8_t* status, uint64_t* tally, uint8_t* c_out) {
    unsigned long long t[4] = {0, 0, 0, 0};
    run(params_vargen(u, R, PK, Gen, m, n, out_ptrs{status, t, c_out, nullptr}));
    if (tally) for (int i = 0; i < 4; ++i) tally[i] = t[i];
    return 0;
}
}
