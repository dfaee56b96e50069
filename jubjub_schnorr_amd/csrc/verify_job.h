// Part of jjs_gpu.hip (included inside its anonymous namespace): one verification call in stages -- the arenas of the
// key-table path, and begin / ingest / keys / hash / finish, which a resident call runs once over all its items and a
// host-buffer call feeds range by range (host_calls.h).
#pragma once
// ---- key-table path: arenas ------------------------------------------------------------------------------
// Batches of at least this many items (big or medium slot) try the key tables: single signatures from 65 536 items on; the
// schemes with two table lookups per item (double signatures, per-item generators), whose throughput path is two equations
// or full-size scalars, from 32 768 (2^15 items under 1 024 keys: double 2.64 -> 2.28 ms, var-gen 2.23 -> 1.96, single
// 1.57 -> 1.70; profiles/r03_quad_chain_ab.txt).
constexpr size_t KT_MIN_ITEMS = 65536, KT_MIN_ITEMS_TWO_LOOKUPS = 32768;
// The table pool a slot starts with: SURVEY.md 8(d)'s 4 096 keys with wide windows are 0.86 GB per column, two columns
// (double, var-gen) 1.73 GB.  A call whose keys repeat but need more leaves a note (key_feedback) and the pool has grown
// by the slot's next call; until then the call runs the throughput path, as it would with keys that do not repeat.
constexpr size_t KEY_POOL_INITIAL_BYTES = size_t(1792) << 20;
size_t pad256(size_t x) { return (x + 255) & ~size_t(255); }

// One column's share of the pool: key_item / flags per key, then bases and tables in the proportion narrow windows
// need (1 : 17); wide windows (1 : 33) then fill 97 % of the same regions.
struct pool_layout {
    size_t item_bytes, flag_bytes, base_bytes, table_bytes;
    uint32_t cap_narrow, cap_wide;      // keys whose bases and tables fit, by window width
};
pool_layout key_pool_layout(size_t col_bytes) {
    pool_layout L{};
    const size_t per_narrow = kt_key_bytes(KT_WINDOW_NARROW) + 8;
    const size_t cap = col_bytes > 4096 ? (col_bytes - 4096) / per_narrow : 0;
    L.cap_narrow = (uint32_t)(cap < 0x7fffffffu ? cap : 0x7fffffffu);
    L.item_bytes = pad256((size_t)L.cap_narrow * 4);
    L.flag_bytes = pad256(L.cap_narrow);
    L.base_bytes = pad256((size_t)L.cap_narrow * kt_positions(KT_WINDOW_NARROW) * KT_BASE_WORDS * 4);
    const size_t used = L.item_bytes + 2 * L.flag_bytes + L.base_bytes;
    L.table_bytes = col_bytes > used ? (col_bytes - used) & ~size_t(255) : 0;
    const size_t by_base = L.base_bytes / ((size_t)kt_positions(KT_WINDOW_WIDE) * KT_BASE_WORDS * 4);
    const size_t by_table = L.table_bytes / ((size_t)kt_positions(KT_WINDOW_WIDE) * kt_table_words(KT_WINDOW_WIDE) * 4);
    L.cap_wide = (uint32_t)(by_base < by_table ? by_base : by_table);
    if ((size_t)L.cap_narrow * kt_positions(KT_WINDOW_NARROW) * kt_table_words(KT_WINDOW_NARROW) * 4 > L.table_bytes) L.cap_narrow = 0;   // cannot happen (see per_narrow)
    return L;
}
// the pool size at which `cols` columns hold `keys` keys each (wide or narrow windows), with some headroom
size_t key_pool_bytes_for(uint32_t cols, uint64_t keys, bool wide) {
    const uint64_t want = keys + keys / 16 + 16;
    size_t col = (size_t)want * (kt_key_bytes(wide ? KT_WINDOW_WIDE : KT_WINDOW_NARROW) + 8) + 8192;
    for (int i = 0; i < 64; ++i) {
        const pool_layout L = key_pool_layout(col);
        if ((wide ? L.cap_wide : L.cap_narrow) >= want) break;
        col += col / 32 + 4096;
    }
    return pad256(col) * cols;
}

// Reads what the slot's previous key-table attempt left in pinned memory (if that call has ended): the path
// statistics, and the pool size a turned-down batch would have needed.
void note_key_feedback() {
    if (!sl->seen_pending || hipEventQuery(sl->last_use) != hipSuccess) return;
    sl->seen_pending = false;
    const uint32_t* c = sl->seen->counters;
    sl->keys_repeated = c[2] != 0;
    if (c[2] == (uint32_t)KT_WINDOW_WIDE) ++g->stats[JJS_PATH_KEY_TABLES_WIDE];
    else if (c[2] == (uint32_t)KT_WINDOW_NARROW) ++g->stats[JJS_PATH_KEY_TABLES_NARROW];
    else if (c[3]) ++g->stats[JJS_PATH_KEYS_PROBE_LIMIT];
    else if (c[4]) {
        ++g->stats[JJS_PATH_KEYS_POOL_TOO_SMALL];
        uint64_t most = 0;
        bool wide = true;
        for (uint32_t k = 0; k < sl->seen_cols && k < 2; ++k) {
            most = c[k] > most ? c[k] : most;
            wide = wide && (uint64_t)c[k] * KT_WIDE_MULTIPLICITY <= sl->seen_n;
        }
        const size_t want = key_pool_bytes_for(sl->seen_cols, most, wide);
        if (want > sl->key_pool_want) sl->key_pool_want = want;
    } else ++g->stats[JJS_PATH_KEYS_DO_NOT_REPEAT];
}

int ensure_key_index(size_t bytes) {
    if (bytes <= sl->keys_bytes) return JJS_OK;
    const size_t cap = grown(bytes);
    return regrow(sl->keys, sl->keys_bytes, sl->keys_bytes, cap, cap);
}
// key columns of a scheme: PK of every equation, and the generator where it is per-item data
uint32_t key_columns(const verify_params& P, fe_src cols[2]) {
    uint32_t n_cols = 0;
    for (uint32_t e = 0; e < P.n_eq; ++e) {
        if (cols) cols[P.eq[e].pk_col] = P.eq[e].pk;
        n_cols = n_cols > (uint32_t)P.eq[e].pk_col + 1 ? n_cols : (uint32_t)P.eq[e].pk_col + 1;
        if (!P.eq[e].comb) {
            if (cols) cols[P.eq[e].gen_col] = P.eq[e].gen;
            n_cols = n_cols > (uint32_t)P.eq[e].gen_col + 1 ? n_cols : (uint32_t)P.eq[e].gen_col + 1;
        }
    }
    return n_cols;
}
// The pool is sized by the call at hand -- no batch takes the path with more than n / 16 keys per column, so a medium slot's
// single-column calls never ask for more than ~1.1 GB -- up to KEY_POOL_INITIAL_BYTES, and beyond that only when a call has
// shown that its keys repeat and need more (key_pool_want).  When hipMalloc says no, the pool the slot has stays (and that
// size is not asked for again).  The replacement exists before the old pool is retired; nobody waits for the device.
int ensure_key_pool(const verify_params& P) {
#if defined(JJS_PROFILING)
    if (g_fail_key_arena) return fail(JJS_ERR_HIP, "key arena allocation failed (jjs_debug_fail_key_arena)");
#endif
    const uint32_t n_cols = key_columns(P, nullptr);
    size_t first = key_pool_bytes_for(n_cols ? n_cols : 1, P.n / KT_MIN_MULTIPLICITY, false);
    if (first > KEY_POOL_INITIAL_BYTES) first = KEY_POOL_INITIAL_BYTES;
    size_t want = sl->key_pool_want > first ? sl->key_pool_want : first;
    const bool refused = sl->key_pool_refused && want >= sl->key_pool_refused;     // hipMalloc has said no to this much before
    if (sl->key_pool && (want <= sl->key_pool_bytes || refused)) return JJS_OK;
    if (!sl->key_pool && refused) want = first < sl->key_pool_refused ? first : sl->key_pool_refused / 2;
    uint8_t* fresh = nullptr;
    if (hipMalloc(&fresh, want) != hipSuccess) {
        (void)hipGetLastError();
        sl->key_pool_refused = want;
        ++g->stats[JJS_PATH_KEYS_NO_MEMORY];
        return sl->key_pool ? JJS_OK : fail(JJS_ERR_HIP, "hipMalloc of the key-table pool (%zu bytes) failed", want);
    }
    retire(sl->key_pool, false, sl->key_pool_bytes);        // earlier launches may still read the old pool
    sl->key_pool = fresh;
    sl->key_pool_bytes = want;
    return JJS_OK;
}

bool key_path_applies(const verify_params& P) {
#if defined(JJS_PROFILING)
    if (g_force_path == 3) return false;           // throughput path without the key tables
#endif
    if (P.n_eq < 1) return false;
    const bool one_lookup = P.n_eq == 1 && P.eq[0].comb != nullptr;          // single signatures: c * PK only
    return P.n >= (one_lookup ? KT_MIN_ITEMS : KT_MIN_ITEMS_TWO_LOOKUPS) && P.n <= 0x7fffffffu && sl->key_stream != nullptr;
}
// The compressed key columns of a wire call: decoded once per key when the key tables engage and once per item
// otherwise, into the affine columns the scheme descriptor already points at.
struct wire_keys {
    uint32_t n_cols = 0;
    fe_src comp[2];          // 32-byte encodings, in the order of the scheme's key columns (eq_desc::pk_col / gen_col)
    uint8_t* out[2] = {};    // n x 64 affine
    uint8_t* bad = nullptr;  // n malformed flags
    decode_params sig{};     // the R points of the signatures (decoded per item, beside the key kernels)
};
uint64_t next_seed() {       // per-call seed of the dedup hash: unpredictable to whoever chose the keys
    static std::mutex mu;    // the per-device workers of a multi-device host call get here concurrently
    static std::mt19937_64 rng = [] {
        std::random_device rd;
        std::seed_seq seq{rd(), rd(), rd(), rd(), (unsigned)std::chrono::steady_clock::now().time_since_epoch().count()};
        return std::mt19937_64(seq);
    }();
    std::lock_guard<std::mutex> lock(mu);
    return rng();
}
// Carves the key buffers of this call out of the slot's two arenas (growing them if need be); what the call must find
// cleared -- counters, cursors, the hash table of every column -- lies side by side at [*cleared, *cleared + *cleared_bytes).
int carve_keys(const verify_params& P, key_params& K, uint8_t** cleared, size_t* cleared_bytes) {
    K.n = P.n;
    fe_src cols[2];
    const uint32_t n_cols = key_columns(P, cols);
    K.n_cols = n_cols;
    if (int rc = ensure_key_pool(P)) return rc;
    const size_t col_bytes = (sl->key_pool_bytes / n_cols) & ~size_t(255);
    const pool_layout L = key_pool_layout(col_bytes);
    const uint64_t most = P.n / KT_MIN_MULTIPLICITY;            // more keys than this never take the path
    K.max_keys = (uint32_t)(L.cap_narrow < most ? L.cap_narrow : most);
    K.max_keys_wide = L.cap_wide;
    if (K.max_keys == 0) return fail(JJS_ERR_HIP, "key-table pool too small");
    size_t slots = 1;
    while (slots < 2 * P.n) slots <<= 1;
    const size_t per_col = pad256(slots * 4) + 2 * pad256(P.n * 4);
    const size_t cursor_words = (size_t)K.max_keys + 1 > (size_t)CURSOR_DENSE_FROM * CURSOR_STRIDE ? (size_t)K.max_keys + 1 : (size_t)CURSOR_DENSE_FROM * CURSOR_STRIDE;
    const size_t order_bytes = pad256(P.n * 4) + pad256(cursor_words * 4);
    const size_t valid_bytes = pad256(((size_t)K.max_keys + 1) * 4);
    if (int rc = ensure_key_index(256 + order_bytes + n_cols * (per_col + valid_bytes))) return rc;
    uint8_t* p = sl->keys;
    // what every call finds cleared comes first and side by side (one memset, one launch: the dozen small dependent launches at
    // the head of a call are a third of a 2^16-item batch)
    *cleared = p;
    K.counters = reinterpret_cast<uint32_t*>(p); p += 256;
    K.key_cursor = reinterpret_cast<uint32_t*>(p); p += pad256(cursor_words * 4);
    for (uint32_t c = 0; c < n_cols; ++c) {
        K.col[c].hash = reinterpret_cast<uint32_t*>(p); K.col[c].hash_mask = (uint32_t)(slots - 1); p += pad256(slots * 4);
    }
    *cleared_bytes = (size_t)(p - *cleared);
    K.order = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
    for (uint32_t c = 0; c < n_cols; ++c) {
        key_column& C = K.col[c];
        C.src = cols[c];
        C.key_bytes = 64;
        C.rep = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
        C.keyid = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
        C.valid_ids = reinterpret_cast<uint32_t*>(p); p += valid_bytes;
        uint8_t* q = sl->key_pool + (size_t)c * col_bytes;
        C.key_item = reinterpret_cast<uint32_t*>(q); q += L.item_bytes;
        C.key_flags = q; q += L.flag_bytes;
        C.key_undecodable = q; q += L.flag_bytes;
        C.bases = reinterpret_cast<uint32_t*>(q); q += L.base_bytes;
        C.tables = reinterpret_cast<uint32_t*>(q);
    }
    return JJS_OK;
}
// The key buffers of this call, and their clearing on `s`.
int setup_keys(const verify_params& P, key_params& K, hipStream_t s) {
    note_key_feedback();
    K.seed = next_seed();
#if defined(JJS_PROFILING)
    K.force_window = (uint32_t)g_force_window;
    K.keep_order = g_keep_order ? 1u : 0u;
    if (g_pin_hash_seed) K.seed = 0;
#endif
    uint8_t* cleared = nullptr;
    size_t cleared_bytes = 0;
    if (int rc = carve_keys(P, K, &cleared, &cleared_bytes)) return rc;
    HIP_TRY(hipMemsetAsync(cleared, 0, cleared_bytes, s));
    return JJS_OK;
}

bool small_path_applies(const verify_params& P) {
    if (P.n_eq < 1 || P.n_eq > 2) return false;
    const bool vargen = P.eq[0].comb == nullptr;        // single: 1 fixed-generator equation, double: 2, var-gen: 1 per-item
    if (vargen && P.n_eq != 1) return false;
    for (uint32_t k = 1; k < P.n_eq; ++k)
        if (!P.eq[k].comb) return false;
#if defined(JJS_PROFILING)
    if (g_force_path == 1 || g_force_path == 3) return false;
    if (g_force_path == 2) return P.n <= MEDIUM_SLOT_ITEMS;
#endif
    return P.n <= (vargen ? SMALL_PATH_MAX_ITEMS_VARGEN : SMALL_PATH_MAX_ITEMS[P.n_eq]);
}

// jjs_reserve: the slot buffers a call with this descriptor would allocate (the slot is `sl`), allocated now
int reserve_for(const verify_params& P) {
    if (int rc = ensure_prep(P.n)) return rc;
    if (small_path_applies(P)) return ensure_small(small_table_bytes_max(P) + 4 * P.n + 64);
    if (int rc = ensure_pending(P.n)) return rc;
    if (key_path_applies(P)) {
        key_params K{};
        uint8_t* cleared = nullptr;
        size_t cleared_bytes = 0;
        if (carve_keys(P, K, &cleared, &cleared_bytes) != JJS_OK) (void)hipGetLastError();     // as in a call: no pool, no key tables
    }
    return JJS_OK;
}

// ---- one verification call, in stages --------------------------------------------------------------------
// A call is: begin (buffers, ordering against the slot's previous user, the key stream forked off), ingest (format
// conversion of columns that have arrived: normalisation of extended points, decoding of the R points of a wire call),
// keys (the key kernels, once every key column is in place), hash (challenge hashes and the other per-item preparation
// of a range of items whose columns are all in place) and finish (the equations, the resolve pass).  A resident call
// runs the stages once over all its items (launch_staged); a host-buffer call feeds them range by range while the
// later ranges are still being uploaded, so that the keys of the whole call are counted and tabled ONCE and the
// hashes start with the first bytes that arrive (run_host_block).
//   throughput path, three launches per batch: prepare (hashes, scalar lattice, subgroup tests; high occupancy), verify
//   (the equations; register-bound) and the resolve pass over the items verify queued (normally the invalid ones only);
//   key-table path: the key kernels on the slot's key stream beside the hashes, then key_verify_kernel; whichever of
//   verify_kernel / key_verify_kernel is not wanted leaves at once;  small batches take the latency path instead.
// which columns of a range have just arrived: the key columns, the other columns the hashes read, or both; COLS_LATE (host-
// buffer calls only) = the columns nothing reads before job_finish (u: the head launch of prepare_kernel does not touch it)
enum : uint32_t { COLS_KEYS = 1, COLS_REST = 2, COLS_ALL = 3, COLS_LATE = 4 };
struct staged_call {
    verify_params P{};
    bool tally_cleared = false;       // the caller's counters arrive zeroed (a lane call uploads them with its inputs)
    bool wire = false;                // compressed points: W
    wire_keys W{};
    bool ext = false;                 // extended points: N[COLS_KEYS] the key columns, N[COLS_REST] the others, N[COLS_ALL] all
    normalize_params N[4] = {};
};
struct verify_job {
    staged_call C;
    key_params K{}, Kd{};
    key_decode_params KD{};
    hipStream_t s = nullptr;          // the caller's stream: begin and finish are queued on it
    hipStream_t side[HOST_SIDE_STREAMS] = {};   // further streams ranges were queued on (host-buffer calls), joined by finish
    bool small = false, try_keys = false, split = false, forked = false, keys_queued = false, open = false;
    bool host_fed = false;            // a host-buffer call: the hashes are queued range by range as the uploads arrive
};

int launch_normalize(normalize_params N, uint64_t first, uint64_t count, uint64_t n_call, hipStream_t s) {
    if (!N.n_src || !count) return JJS_OK;
    N.first = first; N.n = count;
    size_t blocks = (count + BLOCK - 1) / BLOCK;
    const size_t by_share = (count + (size_t)BLOCK * 32 - 1) / ((size_t)BLOCK * 32);
    if (count == n_call) {
        // a whole call: ~8 items per lane at BASELINE sizes (one inversion amortised over them), one item per lane for small calls
        // (a resident call normalises everything ahead of its hashes: 0.25 ms at 2^20 items, 2.7 % of the batch's instructions.
        // Hiding it beside the hashes does not pay -- the chip is bound by instruction issue, so the work costs what it costs
        // wherever it runs: the leading 2^16 items normalised first and the rest on a priority stream beside their hashes
        // measured 1-3 % SLOWER, 16 items per lane instead of 8 the same within noise: profiles/r04_ext_pipeline_ab*.txt)
        if (blocks > 512) blocks = 512;
    } else {
        // a range of a host-buffer call: ~8 items per lane from 2^18 items on (1, 2 or 4 items per lane, and the normalisation
        // done by the hashing lanes themselves instead of a launch in front of them, are the same to a 2^20-item call within
        // noise: such a call is bound by its uploads and by the key kernels behind the last key column, profiles/r04_host_ext_ab.jsonl)
        const size_t few = blocks < 64 ? blocks : 64, shared = (count + (size_t)BLOCK * 8 - 1) / ((size_t)BLOCK * 8);
        blocks = few > shared ? few : shared;
        if (blocks > 512) blocks = 512;
    }
    if (blocks < by_share) blocks = by_share;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, s, N);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
int launch_key_decode_per_item(const verify_job& J, uint64_t first, uint64_t count, const uint32_t* skip_flag, hipStream_t s) {
    const wire_keys& W = J.C.W;
    decode_params D{};
    D.n_src = W.n_cols; D.n = count; D.first = first; D.bad = W.bad;
    for (uint32_t c = 0; c < W.n_cols; ++c) { D.src[c] = W.comp[c]; D.out[c] = W.out[c]; }
    D.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
    D.skip_flag = skip_flag;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)grid_for(8192, count)), dim3(BLOCK), 0, s, D);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}

// The slot has been chosen by the caller (pick_slot) and holds the buffers the descriptor points at.
int job_begin(verify_job& J, hipStream_t s) {
    verify_params& P = J.C.P;
    J.s = s;
#if defined(JJS_PROFILING)
    P.skip_phases = g_skip_phases;
#endif
    if (int rc = ensure_prep(P.n)) return rc;
    P.prep = sl->prep;
    P.workspace = sl->workspace;
    if (int rc = begin_shared(s)) return rc;
    J.open = true;
    clear_params Z{};
    Z.p[0] = P.tally; Z.bytes[0] = P.tally && !J.C.tally_cleared ? 4 * sizeof(unsigned long long) : 0;
    Z.p[1] = const_cast<uint8_t*>(P.pre_malformed); Z.bytes[1] = P.pre_malformed ? P.n : 0;
    J.small = small_path_applies(P);
    if (!J.small) {
        if (int rc = ensure_pending(P.n)) return rc;
        P.pending_count = reinterpret_cast<unsigned long long*>(sl->pending);
        P.pending = sl->pending + 2;
        Z.p[2] = sl->pending; Z.bytes[2] = sizeof(uint64_t);
    }
    if (Z.bytes[0] || Z.bytes[1] || Z.bytes[2]) {
        const uint64_t units = Z.bytes[1] / 16 / BLOCK + 1;
        hipLaunchKernelGGL(clear_kernel, dim3((unsigned)(units < 256 ? units : 256)), dim3(BLOCK), 0, s, Z);
        HIP_TRY(hipGetLastError());
    }
    if (J.small) { ++g->stats[JJS_PATH_LATENCY]; return JJS_OK; }
    J.try_keys = key_path_applies(P);
    if (!J.try_keys) { ++g->stats[JJS_PATH_THROUGHPUT]; return JJS_OK; }
    // The keys are counted (and, for a wire call, decoded once each) on the slot's key stream, with the clearing of
    // their tables, beside the first kernels of the caller's stream.  With affine or extended inputs those are the challenge
    // hashes, which do not wait for the decision (PREP_HEAD; PREP_TAIL later adds what only the throughput path needs); a wire
    // call decodes the R points of its signatures meanwhile and hashes once its keys are in place.
    HIP_TRY(hipEventRecord(sl->key_fork, s));
    HIP_TRY(hipStreamWaitEvent(sl->key_stream, sl->key_fork, 0));
    J.forked = true;
    if (setup_keys(P, J.K, sl->key_stream) != JJS_OK) {
        // no room for the key tables: the batch simply takes the throughput path, as it would with keys that do not repeat
        (void)hipGetLastError();
        J.try_keys = false;
        ++g->stats[JJS_PATH_THROUGHPUT];
        return JJS_OK;
    }
    P.key_flag = J.K.counters + 2;
    // A launch that reads the decision word must find this call's cleared word or this call's decision, never what the slot's
    // previous call left there.  A resident call's readers are ordered behind the key stream anyway (PREP_HEAD does not read the
    // word; a wire call hashes behind key_mid; PREP_TAIL and the equations run behind key_join).  A host-buffer call hashes its
    // first ranges before its key kernels are even queued, and a wire call does so with PREP_ALL, which reads the word: a stale
    // non-zero value would give those items records without half-size scalars and without validated keys.  So its streams --
    // all forked from `s` behind this point -- start behind the clearing (its first upload takes longer than that anyway).
    if (J.host_fed) {
        HIP_TRY(hipEventRecord(sl->key_cleared, sl->key_stream));
        HIP_TRY(hipStreamWaitEvent(s, sl->key_cleared, 0));
    }
#if defined(JJS_AB_NO_SPLIT)        // build-time knob of the A/B run recorded in DESIGN.md 6
    J.split = false;
#else
    J.split = !J.C.wire;
#endif
    return JJS_OK;
}

// format conversion of the columns `cols` of the items [first, first + count), which are now in device memory
int job_ingest(verify_job& J, uint64_t first, uint64_t count, uint32_t cols, hipStream_t cs) {
    if (!count) return JJS_OK;
    if (J.C.ext)
        if (int rc = launch_normalize(J.C.N[cols & 3u], first, count, J.C.P.n, cs)) return rc;
    if (J.C.wire && (cols & COLS_REST) && !J.small) {  // R (R') of every item (the latency path decodes every point of the call
                                                       // in ONE launch, a lane each: job_hash)
        decode_params D = J.C.W.sig;
        D.first = first; D.n = count;
        D.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
        hipLaunchKernelGGL(decode_kernel, dim3((unsigned)grid_for(8192, count)), dim3(BLOCK), 0, cs, D);
        HIP_TRY(hipGetLastError());
    }
    return JJS_OK;
}

#ifndef JJS_KEYS_AHEAD_MIN_ITEMS
#define JJS_KEYS_AHEAD_MIN_ITEMS (3u << 16)
#endif
constexpr uint64_t KEYS_AHEAD_MIN_ITEMS = JJS_KEYS_AHEAD_MIN_ITEMS, KEYS_AHEAD_MAX_ITEMS = 1u << 18;      // see launch_staged
constexpr uint64_t TABLES_BEHIND_MIN_ITEMS = 3u << 18;   // see job_keys
// Every key column of the call is in place (on the key stream's timeline: the caller has made it wait for whatever
// put them there): count the distinct keys, decide on the device, build the per-key tables.
int job_keys(verify_job& J) {
    if (!J.try_keys) return JJS_OK;
    const verify_params& P = J.C.P;
    key_params& K = J.K;
    hipStream_t ks = sl->key_stream;
    const unsigned item_blocks = (unsigned)grid_for(8192, P.n);
    // Four lanes per key for the chains (a third of the latency, twice the instructions) where the batch waits for them: up to
    // 2^18 items, and with two key columns, whose key kernels take twice as long and end after the hashes of a 2^20 batch
    // (var-gen 12.36 -> 11.94 ms, double unchanged; a single 2^20 batch does not wait for its chains and is 1 % slower
    // with them on four lanes: profiles/r03_quad_chain_ab.txt)
#if defined(JJS_AB_CHAIN_ONE_LANE)        // build-time knob of the A/B run recorded in DESIGN.md 6
    K.quad_chains = 0;
#else
    // (nor does a host-fed 2^20 call gain from them, although its key kernels end with its hashes: profiles/r04_host_ext_ab.jsonl)
    K.quad_chains = (P.n <= KEYS_AHEAD_MAX_ITEMS || K.n_cols >= 2) ? 1u : 0u;
#endif
    J.Kd = K;                                   // a wire call deduplicates the 32-byte encodings
    if (J.C.wire)
        for (uint32_t c = 0; c < K.n_cols; ++c) { J.Kd.col[c].src = J.C.W.comp[c]; J.Kd.col[c].key_bytes = 32; }
    hipLaunchKernelGGL(key_dedup_kernel, dim3(item_blocks), dim3(BLOCK), 0, ks, J.Kd);
    hipLaunchKernelGGL(key_assign_kernel, dim3(item_blocks), dim3(BLOCK), 0, ks, J.Kd);
    hipLaunchKernelGGL(key_spread_kernel, dim3(item_blocks), dim3(BLOCK), 0, ks, J.Kd);
    const unsigned key_blocks = (K.n_cols * K.max_keys + BLOCK - 1) / BLOCK;
    if (J.C.wire) {
        // one square root per distinct key; the items fetch their key's point in job_hash
        for (uint32_t c = 0; c < K.n_cols; ++c) J.KD.out[c] = J.C.W.out[c];
        J.KD.bad = J.C.W.bad;
        J.KD.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
        hipLaunchKernelGGL(key_decode_kernel, dim3(key_blocks), dim3(BLOCK), 0, ks, J.Kd, J.KD);
        HIP_TRY(hipEventRecord(sl->key_mid, ks));              // the decision and the decoded keys
    }
    hipLaunchKernelGGL(key_count_kernel, dim3(item_blocks), dim3(BLOCK), 0, ks, K);
    hipLaunchKernelGGL(key_scan_kernel, dim3(1), dim3(1024), 0, ks, K);
    hipLaunchKernelGGL(key_scatter_kernel, dim3(item_blocks), dim3(BLOCK), 0, ks, K);
    HIP_TRY(hipEventRecord(sl->key_ahead, ks));                // the keys are counted and grouped: see launch_staged
    hipLaunchKernelGGL(key_chain_kernel, dim3(((K.quad_chains ? 5 : 1) * K.n_cols * K.max_keys + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, ks, K);
    // The tables are throughput work (2 700 waves of 50 k instructions for 4 096 keys).  In a batch whose hashes outlast the
    // key kernels nothing needs them before the hashes have ended: on a stream of the LOWEST priority their blocks are
    // dispatched when prepare_kernel has no block left to dispatch, into the wave slots its last blocks leave idle.  On the key
    // stream (high priority, for the sake of the chains) they displaced hash waves in the middle of the batch: 2^20 items 9.67 ->
    // 9.07 ms single, 16.95 -> 15.85 double, 11.96 -> 11.25 var-gen.  A smaller batch waits for its tables, and they stay on
    // the key stream (2^18 items: 2.95 ms there, 3.43 behind the hashes; 2^19: 5.93 / 6.02; profiles/r03_table_stream_ab.txt).
#if defined(JJS_AB_TABLES_ON_KEY_STREAM)     // build-time knobs of that A/B run
    hipStream_t ts = ks;
#elif defined(JJS_AB_TABLES_ON_TABLE_STREAM)
    hipStream_t ts = sl->table_stream;
#else
    // (a host-buffer call of single signatures is still uploading when its key kernels start, and its hashes end with its
    // tables: those stay on the key stream, 10.3-10.8 against 10.7-10.9 ms; double and var-gen calls, whose hashes last longer,
    // gain 0.4-0.8 ms behind them)
    hipStream_t ts = P.n >= TABLES_BEHIND_MIN_ITEMS && !(J.host_fed && K.n_cols == 1) ? sl->table_stream : ks;
#endif
    if (ts != ks) {
        HIP_TRY(hipEventRecord(sl->key_chains, ks));
        HIP_TRY(hipStreamWaitEvent(ts, sl->key_chains, 0));
    }
    hipLaunchKernelGGL(key_table_kernel, dim3((unsigned)(((uint64_t)K.n_cols * K.max_keys * KT_MAX_POSITIONS + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, ts, K);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl->key_join, ts));
    J.keys_queued = true;
    return JJS_OK;
}

// The items [first, first + count) have all their columns in place and ingested: hash them.
int job_hash(verify_job& J, uint64_t first, uint64_t count, hipStream_t cs) {
    if (!count) return JJS_OK;
    const verify_params& P = J.C.P;
    if (J.small) {
        if (first != 0 || count != P.n) return fail(JJS_ERR_ARG, "internal: the latency path takes the call whole");
        if (J.C.wire) {
            // every compressed point of the call -- the R points of the signatures and the key columns -- in one launch, one lane
            // per point: the call waits for ONE square root (0.17 ms) instead of one per point of an item, one after the other
            // (single 0.33 ms, double 0.70: profiles/r04_wire_small_ab.jsonl)
            const wire_keys& W = J.C.W;
            decode_params D = W.sig;
            for (uint32_t c = 0; c < W.n_cols && D.n_src < 4; ++c) { D.src[D.n_src] = W.comp[c]; D.out[D.n_src] = W.out[c]; ++D.n_src; }
            D.first = 0; D.n = P.n; D.bad = W.bad; D.ok = nullptr; D.split = 1; D.skip_flag = nullptr;
            D.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
            hipLaunchKernelGGL(decode_points_kernel, dim3((unsigned)grid_for(8192, P.n * D.n_src)), dim3(BLOCK), 0, cs, D);
            HIP_TRY(hipGetLastError());
        }
        return launch_small(P, cs);
    }
    if (J.C.wire) {
        if (J.try_keys && J.keys_queued) {
            HIP_TRY(hipStreamWaitEvent(cs, sl->key_mid, 0));
            // this stream decodes the key columns item by item only if the batch turned the key tables down; else every
            // item fetches its key's point
            if (int rc = launch_key_decode_per_item(J, first, count, J.K.counters + 2, cs)) return rc;
            hipLaunchKernelGGL(key_unpack_kernel, dim3((unsigned)grid_for(8192, count)), dim3(BLOCK), 0, cs, J.Kd, J.KD, first, count);
        } else if (int rc = launch_key_decode_per_item(J, first, count, nullptr, cs)) return rc;
    }
    hipLaunchKernelGGL(prepare_kernel, dim3(grid_for(g->grid_prepare, count)), dim3(BLOCK), 0, cs, P, J.split ? (int)PREP_HEAD : (int)PREP_ALL,
                       first, count);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}

int job_finish(verify_job& J) {
    const verify_params& P = J.C.P;
    hipStream_t s = J.s;
    for (hipStream_t& side : J.side) {              // ranges were queued on other streams: they join here
        if (!side) continue;
        HIP_TRY(hipEventRecord(g->side_join, side));
        HIP_TRY(hipStreamWaitEvent(s, g->side_join, 0));
        side = nullptr;
    }
    if (J.small) { J.open = false; return end_shared(s); }
    if (J.try_keys) {
        if (!J.keys_queued) return fail(JJS_ERR_ARG, "internal: finish before the key kernels");
        HIP_TRY(hipStreamWaitEvent(s, sl->key_join, 0));
        J.forked = false;
        if (J.split) hipLaunchKernelGGL(prepare_kernel, dim3(grid_for(g->grid_prepare, P.n)), dim3(BLOCK), 0, s, P, (int)PREP_TAIL, (uint64_t)0, P.n);
        hipLaunchKernelGGL(key_verify_kernel, dim3(grid_for(g->grid_key_verify, P.n)), dim3(BLOCK), 0, s, P, J.K);
    }
    hipLaunchKernelGGL(verify_kernel, dim3(grid_for(sl->grid_verify, P.n)), dim3(BLOCK), 0, s, P);
    hipLaunchKernelGGL(resolve_kernel, dim3(grid_for(g->grid_resolve, P.n * P.resolve_lanes)), dim3(BLOCK), 0, s, P);
    HIP_TRY(hipGetLastError());
    if (J.try_keys) {
        // what the keys of this call looked like, for the slot's next call (note_key_feedback); nobody waits for it
        HIP_TRY(hipMemcpyAsync(sl->seen->counters, J.K.counters, sizeof(sl->seen->counters), hipMemcpyDeviceToHost, s));
        sl->seen_pending = true; sl->seen_n = P.n; sl->seen_cols = J.K.n_cols;
    }
    J.open = false;
    return end_shared(s);
}
// A stage failed: whatever has been queued on the key stream or the second stream still uses the slot's buffers, so
// the caller's stream joins both and the slot's event covers them (the error itself goes back to the caller).
void job_abandon(verify_job& J) {
    if (!J.open) return;
    (void)hipGetLastError();
    if (J.forked) {        // the key stream, and the table stream behind it
        if (hipEventRecord(sl->key_chains, sl->key_stream) == hipSuccess && hipStreamWaitEvent(sl->table_stream, sl->key_chains, 0) == hipSuccess &&
            hipEventRecord(sl->key_join, sl->table_stream) == hipSuccess)
            (void)hipStreamWaitEvent(J.s, sl->key_join, 0);
    }
    for (hipStream_t side : J.side)
        if (side && hipEventRecord(g->side_join, side) == hipSuccess) (void)hipStreamWaitEvent(J.s, g->side_join, 0);
    (void)hipEventRecord(sl->last_use, J.s);
    J.open = false;
}

// A resident call: every stage once, over all items, in the order that puts the key kernels in front of the hashes.
int launch_staged(const staged_call& C, hipStream_t s) {
    if (C.P.n == 0) return JJS_OK;
    verify_job J;
    J.C = C;
    int rc = job_begin(J, s);
    if (!rc) rc = job_ingest(J, 0, C.P.n, COLS_ALL, s);
    if (!rc && J.forked && C.ext) {                 // the key kernels read normalised key columns
        rc = hipEventRecord(g->ingest_done, s) == hipSuccess && hipStreamWaitEvent(sl->key_stream, g->ingest_done, 0) == hipSuccess
                 ? JJS_OK : fail(JJS_ERR_HIP, "event between the caller's stream and the key stream");
    }
    if (!rc) rc = job_keys(J);
    // A batch that fills every wave slot with its hashes but does not outlast its key kernels -- more than 3 * 2^16 items, up
    // to 2^18: one generation of prepare_kernel's blocks -- hashes only once its keys are counted and grouped (0.13-0.22 ms on
    // the empty chip), so that the per-key chains start with the hashes.  Launched beside them, the key kernels waited for the
    // first wave slot to come free (the blocks of prepare_kernel hold theirs for 1.5 ms) and the tables were ready 1.0 ms after
    // the hashes: 2^18 items 3.55 -> 3.05 ms single, 5.6 -> 4.9 double, 4.5 -> 3.85 var-gen (profiles/r03_keys_ahead_ab.txt,
    // r03_timeline_medium.txt).  A smaller batch leaves wave slots free and its key kernels find them at once: waiting costs it
    // 0.1 ms (2^16 items 1.73 -> 1.61 ms, 2^17 2.33 -> 2.22 without).  At 2^19 items the hashes outlast the key kernels either
    // way and double batches lose 0.4 ms by waiting: larger batches start at once.  A batch whose keys turn out not to repeat
    // has waited for nothing (0.25 ms at 2^18): the slot remembers how its last attempt that has ended came out, and after one
    // that built no tables the hashes start at once (a caller who queues batch after batch without waiting for any gives the
    // slot nothing to remember).
#if !defined(JJS_AB_NO_KEYS_AHEAD)        // build-time knob of the A/B run recorded in DESIGN.md 6
    if (!rc && J.try_keys && J.keys_queued && C.P.n > KEYS_AHEAD_MIN_ITEMS && C.P.n <= KEYS_AHEAD_MAX_ITEMS && sl->keys_repeated)
        rc = hipStreamWaitEvent(s, sl->key_ahead, 0) == hipSuccess ? JJS_OK : fail(JJS_ERR_HIP, "event between the key stream and the caller's stream");
#endif
    if (!rc) rc = job_hash(J, 0, C.P.n, s);
    if (!rc) rc = job_finish(J);
    if (rc) job_abandon(J);
    return rc;
}
