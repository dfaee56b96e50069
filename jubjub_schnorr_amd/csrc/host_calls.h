// Part of jjs_gpu.hip (included inside its anonymous namespace): the blocking host-buffer entry points -- upload plan,
// staging copies, and the pipeline that feeds one verification call per device while its uploads run.
#pragma once
// Host-buffer calls.  The batch is cut into one contiguous block of ceil(n / devices) items per driven
// device (the rule of jubjub_schnorr_amd/sharding.py) and every block is driven by its OWN host thread, so that
// the uploads of different devices overlap (one thread issuing pageable copies for all devices would stage them
// one after the other).  A block is ONE verification call on its device (verify_job), fed piece by piece: the device's
// staging threads copy a piece of the caller's (pageable) arrays into one of three pinned staging slots (a single thread
// moves ~11 GB/s), one piece ahead of the block's thread, which queues the piece's upload on the device's copy stream and,
// behind the upload, whatever the piece makes possible: format conversion of its columns, the key kernels once every key
// has arrived, the challenge hashes of the items whose hash inputs are now complete.  The pieces of a block that may take
// the key tables come in this order (plan_pieces): the hash inputs of the first items (so that the hashes start at once),
// in growing pieces up to half the block; the KEY columns of all the others (the keys of the whole call are counted and
// tabled once, beside the hashes); their remaining hash inputs in growing ranges; and last the columns nothing reads
// before the equations (u).  The equations run once at the end, over the whole block, as in a resident call.  Device
// arena, pinned staging and events are per device and only grow.  The tallies are summed over the devices with one RCCL
// all-reduce.  A failing block drains its streams before it reports, so nothing is in flight into the caller's or the
// library's buffers when the call returns an error.
struct host_col { const uint8_t* p; size_t width; uint32_t group; };      // group: COLS_KEYS, COLS_REST or COLS_LATE
// build-time knobs of the A/B runs recorded in DESIGN.md 6 (scripts/host_ab.sh)
#ifndef JJS_HOST_LEAD_LOG2
#define JJS_HOST_LEAD_LOG2 16            // items of the first piece (all columns) ...
#endif
#ifndef JJS_HOST_LEAD_SHARE_DEN
#define JJS_HOST_LEAD_SHARE_NUM 1        // ... more of them, each twice its predecessor, while they stay within NUM/DEN of the block
#define JJS_HOST_LEAD_SHARE_DEN 2
#endif
#ifndef JJS_HOST_REST_LOG2_FIRST
#define JJS_HOST_REST_LOG2_FIRST 17      // items of the first range of remaining columns ...
#endif
#ifndef JJS_HOST_REST_GROWTH
#define JJS_HOST_REST_GROWTH 2           // ... each later one this many times its predecessor ...
#endif
#ifndef JJS_HOST_REST_LOG2_MAX
#define JJS_HOST_REST_LOG2_MAX 18        // ... up to this many
#endif
#ifndef JJS_HOST_KEYS_LOG2_MAX
#define JJS_HOST_KEYS_LOG2_MAX 19        // the largest piece of key columns
#endif
constexpr size_t HOST_LEAD_ITEMS = size_t(1) << JJS_HOST_LEAD_LOG2, HOST_LEAD_SHARE_NUM = JJS_HOST_LEAD_SHARE_NUM, HOST_LEAD_SHARE_DEN = JJS_HOST_LEAD_SHARE_DEN,
                 HOST_REST_ITEMS_FIRST = size_t(1) << JJS_HOST_REST_LOG2_FIRST, HOST_REST_GROWTH = JJS_HOST_REST_GROWTH,
                 HOST_REST_ITEMS_MAX = size_t(1) << JJS_HOST_REST_LOG2_MAX, HOST_KEYS_ITEMS_MAX = size_t(1) << JJS_HOST_KEYS_LOG2_MAX;
// pinned staging slots of a block.  Three, so that the staging copy of piece i + 2 can run while piece i is on the bus and
// piece i + 1 waits for it (eight threads stage at about the speed of the bus: with two slots they took turns).
// A piece travels as one copy per column: a single copy into a landing area, spread over the columns by a kernel, was
// built and measured slower -- that kernel waits up to 0.85 ms for a wave slot once the hashes fill the chip.
constexpr size_t HOST_SLOTS = 3;
#ifndef JJS_HOST_STAGING_THREADS
#define JJS_HOST_STAGING_THREADS 8
#endif
constexpr unsigned HOST_STAGING_THREADS_MAX = JJS_HOST_STAGING_THREADS;
constexpr size_t HOST_STAGING_MIN_BYTES = size_t(4) << 20;     // below this a piece is copied by the calling thread alone

int ensure_stage(size_t bytes) {
    if (bytes <= g->stage_bytes) return JJS_OK;
    const size_t cap = grown(bytes);
    return regrow(g->stage, g->stage_bytes, g->stage_bytes, cap, cap);
}
int ensure_pinned(size_t bytes) {
    if (bytes <= g->pinned_bytes) return JJS_OK;
    const size_t cap = grown(bytes);
    uint8_t* fresh = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&fresh), cap, hipHostMallocDefault));
    retire(g->pinned, true, g->pinned_bytes);
    g->pinned = fresh;
    g->pinned_bytes = cap;
    return JJS_OK;
}

// nothing may leave an extern "C" entry point by exception: the host-buffer calls allocate (block and piece lists)
template <typename F>
int no_throw(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        return fail(JJS_ERR_HIP, "host-side failure: %s", e.what());
    } catch (...) {
        return fail(JJS_ERR_HIP, "host-side failure");
    }
}
struct host_piece {
    size_t first, count;
    uint32_t cols;             // the column groups it carries (COLS_*)
};
struct host_block {
    size_t lo = 0, hi = 0;
    std::vector<host_piece> pieces;       // the plan in use
    std::vector<host_piece> plans[2];     // [0] the late columns travel with the others, [1] they travel last (split calls)
    size_t largest_bytes = 0;  // of a piece in the pinned staging slots, over both plans
    unsigned staging_threads = 1;
    int rc = JJS_OK;
    char err[512] = "";
    unsigned long long tally[4] = {0, 0, 0, 0};
    bool unlocked = false;     // the call runs outside the engine's mutex (one device; see run_host)
    call_slot* owned = nullptr;// ... and owns this slot until it returns
};

// The upload order of a block of nl items.  row_keys / row_rest: bytes per item of the two column groups.
//   * a block that cannot take the key tables: every column of growing ranges of items;
//   * else the key columns travel ahead of the others, but only for the second half of the block.  The bus delivers a
//     2^20-item single batch in 3.7 ms and the chip hashes it in 5.7: whatever is uploaded ahead of complete items leaves
//     the hashes without input for that long (scripts/host_timeline.sh; keys of the whole block first: the chip idles from
//     1.0 to 2.3 ms), while the per-key tables need ~3.8 ms from the moment the last key has arrived (the doubling chains are
//     latency-bound) and are wanted when the hashes end.  So: all columns of 2^16, 2^17, 2^18 ... items while that stays
//     within half the block, then the key columns of the rest, then its remaining columns in ranges of 2^17, 2^18, 2^18 ...
//     items (ranges of equal size keep the staging copy of the next range shorter than the upload of this one);
//   * a wire call (wire_points: R points per signature) decodes each distinct key once, which takes the whole key column.
//     With one R point per signature the plan is the one above, and the ranges that travel ahead of the rest's key columns
//     decode the keys of their own items instead (a square root per item: 0.75 ms of arithmetic for the 7/16 of a 2^20-item
//     block that they are, done while the chip would wait for the bus).  With two (double signatures: 4 ms of square roots,
//     more than their upload takes) the key columns travel LAST, whole: the R points are decoded at the pace of the bus with
//     nothing else on the chip, and every range is hashed behind the key kernels' decoding.  Measured, single / double calls of
//     2^20 items: key columns first 13.8 / 24.0 ms, last 13.7 / 21.7, the plan above 13.4 / 24.3 (profiles/r03_host_ab_wire_plan.jsonl).
//   * the columns nothing reads before the equations (u) travel last, behind everything the hashes need, when the call
//     hashes with the head launch (`late`: 16 % fewer bytes ahead of the first hashes of a single batch); else with the others.
// row_keys / row_rest / row_late: bytes per item of the column groups.
void plan_pieces(std::vector<host_piece>& pieces, size_t& largest_bytes, size_t nl, size_t row_keys, size_t row_rest, size_t row_late,
                 int wire_points, bool late) {
    pieces.clear();
    const uint32_t rest = COLS_REST | (late ? 0u : COLS_LATE);       // the groups that travel as "the other columns"
    auto add = [&](size_t first, size_t count, uint32_t cols) {
        if (!count) return;
        pieces.push_back(host_piece{first, count, cols});
        const size_t bytes = count * ((cols & COLS_KEYS ? row_keys : 0) + (cols & COLS_REST ? row_rest : 0) + (cols & COLS_LATE ? row_late : 0));
        if (bytes > largest_bytes) largest_bytes = bytes;
    };
    // ranges: `first_len`, then times `growth` up to `cap`; a remainder of less than half a first range joins the range before it
    auto ranges = [&](size_t from, size_t first_len, size_t growth, size_t cap, uint32_t cols) {
        size_t pos = from, next = first_len < cap ? first_len : cap;
        while (pos < nl) {
            size_t len = next < nl - pos ? next : nl - pos;
            if (nl - pos - len < first_len / 2) len = nl - pos;
            add(pos, len, cols);
            pos += len;
            next = next * growth < cap ? next * growth : cap;
        }
    };
    // very large blocks: larger pieces, so that their number stays within the events a device has
    size_t cap_keys = HOST_KEYS_ITEMS_MAX, cap_rest = HOST_REST_ITEMS_MAX;
    if (nl > cap_keys * 8) cap_keys = ((nl + 7) / 8 + 255) & ~size_t(255);
    if (nl > cap_rest * 16) cap_rest = ((nl + 15) / 16 + 255) & ~size_t(255);
    const bool keys_first = row_keys != 0 && nl >= KT_MIN_ITEMS && nl > 2 * HOST_LEAD_ITEMS;
    if (!keys_first) {
        ranges(0, HOST_LEAD_ITEMS, 4, cap_rest, COLS_KEYS | rest);
    } else {
        bool keys_last = wire_points >= 2;
#if defined(JJS_HOST_WIRE_KEYS_LAST)       // build-time knobs of the A/B run recorded in DESIGN.md 6
        keys_last = wire_points >= 1;
#elif defined(JJS_HOST_WIRE_KEYS_LEAD) || defined(JJS_HOST_WIRE_KEYS_FIRST)
        keys_last = false;
#endif
        if (keys_last) {
            ranges(0, HOST_REST_ITEMS_FIRST, HOST_REST_GROWTH, cap_rest, rest);
            ranges(0, cap_keys, 1, cap_keys, COLS_KEYS);
#if defined(JJS_HOST_WIRE_KEYS_FIRST)
        } else if (wire_points) {
            ranges(0, cap_keys, 1, cap_keys, COLS_KEYS);
            ranges(0, HOST_REST_ITEMS_FIRST, HOST_REST_GROWTH, cap_rest, rest);
#endif
        } else {
            size_t lead = 0, next = HOST_LEAD_ITEMS;
            do {
                add(lead, next, COLS_KEYS | rest);
                lead += next;
                next = next * 2 < cap_rest ? next * 2 : cap_rest;
            } while (lead + next <= nl / HOST_LEAD_SHARE_DEN * HOST_LEAD_SHARE_NUM);
            ranges(lead, cap_keys, 1, cap_keys, COLS_KEYS);
            ranges(lead, HOST_REST_ITEMS_FIRST, HOST_REST_GROWTH, cap_rest, rest);
        }
    }
    if (late && row_late) ranges(0, cap_keys, 1, cap_keys, COLS_LATE);
}

struct stage_task {          // one piece's pageable -> pinned copy, cut into T slices of every column
    const host_col* cols; size_t n_cols;
    size_t lo, first, count;
    uint32_t group;
    uint8_t* dst;
    unsigned T;
};
// pageable -> pinned with streaming stores: the pinned slot is written once and read by the DMA engine only, so the
// lines need not be fetched before they are written nor kept in the cache afterwards (memcpy does both for copies of this
// size per thread).  dst 32-byte aligned; falls back to memcpy on a host without AVX2.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(JJS_HOST_NO_STREAM_COPY)
__attribute__((target("avx2"))) void stream_copy_avx2(uint8_t* dst, const uint8_t* src, size_t bytes) {
    size_t i = 0;
    for (; i + 128 <= bytes; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < bytes) memcpy(dst + i, src + i, bytes - i);
}
void stream_copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0 && bytes >= 4096) stream_copy_avx2(dst, src, bytes);
    else memcpy(dst, src, bytes);
}
#else
void stream_copy(uint8_t* dst, const uint8_t* src, size_t bytes) { memcpy(dst, src, bytes); }
#endif
void stage_slice(void* ctx, unsigned t) {
    const stage_task& S = *static_cast<const stage_task*>(ctx);
    const size_t i0 = S.count * t / S.T, i1 = S.count * (t + 1) / S.T;
    uint8_t* q = S.dst;
    for (size_t k = 0; k < S.n_cols; ++k) {
        if (!(S.cols[k].group & S.group)) continue;
        const size_t w = S.cols[k].width;
        stream_copy(q + i0 * w, S.cols[k].p + (S.lo + S.first + i0) * w, (i1 - i0) * w);
        q += S.count * w;
    }
}

// Builds the call of one block from its device arrays (cols[k] of the block at dev[k]; nl items; statuses to st,
// counters to tl): what the *_locked functions below do for a resident call, minus the launch.
typedef int (*call_builder)(const void* const* dev, size_t nl, void* st, void* tl, hipStream_t s, staged_call& out);

// The pipeline of one device's block; runs on the calling thread (one device) or on a thread of its own.
int run_host_block(device_state* dev, const host_col* cols, size_t n_cols, host_block& b, uint8_t* status, call_builder build,
                   verify_job& J) {
    g = dev;
#if defined(JJS_PROFILING)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_stage = 0, t_wait = 0, t_first = 0;
#endif
    HIP_TRY(hipSetDevice(g->device));
    const size_t nl = b.hi - b.lo;
    if (!nl) {                                           // an empty block still reports (zero) counters
        HIP_TRY(hipStreamWaitEvent(g->stream, g->last_use, 0));
        HIP_TRY(hipMemsetAsync(g->tally, 0, 4 * sizeof(unsigned long long), g->stream));
        HIP_TRY(hipEventRecord(g->last_use, g->stream));
        return JJS_OK;
    }
    // device arena: one array per column for the whole block, then the statuses;
    // pinned staging: HOST_SLOTS slots of one piece each, then the statuses of the whole block
    const size_t slot_bytes = pad256(b.largest_bytes);
    size_t bytes = 0;
    for (size_t k = 0; k < n_cols; ++k) bytes += pad256(nl * cols[k].width);
    bytes += pad256(nl);
    if (int rc = ensure_stage(bytes)) return rc;
    const void* in[8];
    uint8_t* col_dev[8];
    uint8_t* p = g->stage;
    for (size_t k = 0; k < n_cols; ++k) { in[k] = col_dev[k] = p; p += pad256(nl * cols[k].width); }
    uint8_t* st = p;
    if (int rc = ensure_pinned(HOST_SLOTS * slot_bytes + pad256(nl) + 256)) return rc;
    uint8_t* const pst = g->pinned + HOST_SLOTS * slot_bytes;
    unsigned long long* const ptally = reinterpret_cast<unsigned long long*>(pst + pad256(nl));
    if (!g->stagers && b.staging_threads > 1) g->stagers = new (std::nothrow) staging_pool(b.staging_threads - 1);
    // The call of this block (the builder picks its slot), and with it the upload order: the columns nothing reads before
    // the equations (u) travel last when the call will hash with the head launch of prepare_kernel, which does not touch
    // them -- i.e. when it tries the key tables (and is not a wire call, whose u sits inside the signature column).
    {
        // the builder picks the call slot (the engine's state): under the engine's mutex when this call does not hold it anyway
        std::unique_lock<std::mutex> lock(L.mu, std::defer_lock);
        if (b.unlocked) lock.lock();
        if (int rc = build(in, nl, st, g->tally, g->stream, J.C)) return rc;
        if (b.unlocked) { sl->host_owned = true; b.owned = sl; }
    }
    bool late = false;
    for (size_t k = 0; k < n_cols; ++k) late = late || cols[k].group == COLS_LATE;
    late = late && !J.C.wire && !small_path_applies(J.C.P) && key_path_applies(J.C.P) && ensure_key_pool(J.C.P) == JJS_OK;
#if defined(JJS_HOST_NO_LATE)            // build-time knob of the A/B run recorded in DESIGN.md 6
    late = false;
#endif
    b.pieces = b.plans[late ? 1 : 0];
    // The staging copy of piece i + 1 runs on the helper threads while this thread queues the uploads and the kernels of
    // piece i (some 0.1 ms of HIP calls per piece, during which the bus would otherwise wait for the next piece).
    stage_task tasks[HOST_SLOTS];
    struct in_flight {          // the helpers read tasks[]: whatever way this function is left, they have finished first
        staging_pool* pool = nullptr;
        bool active = false;
        ~in_flight() { if (active && pool) pool->join(); }
    } staging;
    staging.pool = g->stagers;
    auto stage_begin = [&](size_t i) -> int {
        const host_piece& pc = b.pieces[i];
        if (i >= HOST_SLOTS) HIP_TRY(hipEventSynchronize(g->chunk_up[i - HOST_SLOTS]));      // the slot's previous upload has left it
        size_t piece_bytes = 0;
        for (size_t k = 0; k < n_cols; ++k)
            if (cols[k].group & pc.cols) piece_bytes += pc.count * cols[k].width;
        stage_task& S = tasks[i % HOST_SLOTS];
        S = stage_task{cols, n_cols, b.lo, pc.first, pc.count, pc.cols, g->pinned + (i % HOST_SLOTS) * slot_bytes, 1};
        const unsigned threads = g->stagers ? g->stagers->helpers() + 1 : 1u;
        // slices of about a megabyte, so that whoever is free (helpers, and this thread once it has queued the piece
        // before) takes the next one
        size_t slices = piece_bytes / (size_t(1) << 20);
        if (slices > 4 * (size_t)threads) slices = 4 * (size_t)threads;
        S.T = (piece_bytes >= HOST_STAGING_MIN_BYTES && threads > 1 && slices > 1) ? (unsigned)slices : 1u;
        if (S.T > 1) { g->stagers->begin(S.T, stage_slice, &S); staging.active = true; }
        return JJS_OK;
    };
    auto stage_finish = [&](size_t i) {
        stage_task& S = tasks[i % HOST_SLOTS];
        if (S.T > 1) { g->stagers->join(); staging.active = false; } else stage_slice(&S, 0);
    };
    const size_t np = b.pieces.size();
    if (np > HOST_MAX_PIECES) return fail(JJS_ERR_ARG, "internal: %zu pieces", np);
    if (int rc = stage_begin(0)) return rc;          // ... and of the first piece while this thread sets the call up
    // the arena and the counters may still be in use by the previous call's last launches
    HIP_TRY(hipStreamWaitEvent(g->copy_stream, g->last_use, 0));
    HIP_TRY(hipStreamWaitEvent(g->stream, g->last_use, 0));
    J.host_fed = true;
    if (int rc = job_begin(J, g->stream)) return rc;
    // The ranges of the block are queued on several streams in turn: a launch waits for every block of its predecessor on
    // the same stream, and a block of hashes lives for 1.4 ms, so on one stream the chip runs half empty at the end of
    // every range; with a few streams, whichever range has arrived fills the wave slots that come free.  Three of them
    // (this one and two more): for affine and wire inputs two to six measure the same, for extended inputs three are 0.7 ms
    // ahead of four (profiles/r03_host_ab_streams*.jsonl).  The other streams start behind this one's job_begin (cleared
    // flags and counters).
    HIP_TRY(hipEventRecord(g->host_begin, g->stream));
    hipStream_t compute[1 + HOST_SIDE_STREAMS] = {g->stream};
    for (int k = 0; k < HOST_SIDE_STREAMS; ++k) {
        HIP_TRY(hipStreamWaitEvent(g->side[k], g->host_begin, 0));
        compute[1 + k] = g->side[k];
    }
    constexpr size_t NCS = 1 + HOST_SIDE_STREAMS;
    size_t hash_items[NCS] = {};                        // items whose hashes have been queued on compute[k]
    hipStream_t converted_on[HOST_MAX_PIECES] = {};     // the stream a piece's columns were converted on (job_ingest)
    size_t last_key_piece = np;                         // the piece whose arrival completes the key columns
    for (size_t i = 0; i < np; ++i)
        if (b.pieces[i].cols & COLS_KEYS) last_key_piece = i;
    struct deferred { size_t first, count; hipStream_t cs; };
    // ranges whose hashes cannot be queued yet: a call that was expected to hash with the head launch but does not after
    // all (job_begin could not set the key tables up) reads u, which then travels last; and ranges whose key columns travel
    // behind them (wire calls of double signatures)
    std::vector<deferred> waiting;
    const bool needs_late = late && !J.split;
    size_t last_late_piece = np;
    for (size_t i = 0; i < np; ++i)
        if (b.pieces[i].cols & COLS_LATE) last_late_piece = i;
    // key columns that travel behind the other columns of their items (wire calls): nothing of those items is hashed before
    bool keys_trail = false;
    for (size_t i = 0; i < np; ++i)
        for (size_t j = i + 1; j < np; ++j) {
            const host_piece &a = b.pieces[i], &k = b.pieces[j];
            keys_trail = keys_trail || ((a.cols & COLS_REST) && (k.cols & COLS_KEYS) && k.first < a.first + a.count && a.first < k.first + k.count);
        }
    auto hashes_blocked = [&](size_t i) {
        return (needs_late && i < last_late_piece) || (keys_trail && i < last_key_piece);
    };
    for (size_t i = 0; i < np; ++i) {
        const host_piece& pc = b.pieces[i];
        uint8_t* const hp = g->pinned + (i % HOST_SLOTS) * slot_bytes;
#if defined(JJS_PROFILING)
        const double t0 = now();
#endif
        stage_finish(i);
#if defined(JJS_PROFILING)
        const double t1 = now();
        t_stage += t1 - t0;
#endif
        if (i + 1 < np)
            if (int rc = stage_begin(i + 1)) return rc;
#if defined(JJS_PROFILING)
        t_wait += now() - t1;
#endif
        {
            uint8_t* q = hp;
            for (size_t k = 0; k < n_cols; ++k) {
                if (!(cols[k].group & pc.cols)) continue;
                const size_t w = cols[k].width;
                HIP_TRY(hipMemcpyAsync(col_dev[k] + pc.first * w, q, pc.count * w, hipMemcpyHostToDevice, g->copy_stream));
                q += pc.count * w;
            }
        }
        HIP_TRY(hipEventRecord(g->chunk_up[i], g->copy_stream));
#if defined(JJS_PROFILING)
        if (i == 0) t_first = now() - t_begin;
#endif
        // the stream with the fewest items to hash queued on it so far (in turn, they leave a short last range behind the
        // longest one: a wire call's last 2^17 items hashed alone for 0.8 ms, scripts/host_timeline.sh)
        size_t pick = 0;
        for (size_t k = 1; k < NCS; ++k)
            if (hash_items[k] < hash_items[pick]) pick = k;
        if (pc.cols & COLS_REST) hash_items[pick] += pc.count;
        hipStream_t cs = compute[pick];
        if (pick) J.side[pick - 1] = cs;
        // Extended points are normalised on a stream of higher priority than the hashes: that kernel is a few waves with a
        // long dependent chain (one inversion per lane), its piece cannot be hashed before it ends, and behind the hashes of
        // the pieces before it it waited 0.8-1.2 ms for wave slots instead of running 0.25 (scripts/host_timeline.sh).
        hipStream_t is = J.C.ext ? g->ingest[i & 1] : cs;
        converted_on[i] = is;
        HIP_TRY(hipStreamWaitEvent(is, g->chunk_up[i], 0));
        if (is != cs && i < 2) HIP_TRY(hipStreamWaitEvent(is, g->host_begin, 0));     // behind job_begin's cleared flags
        if (int rc = job_ingest(J, pc.first, pc.count, pc.cols & COLS_ALL, is)) return rc;
        HIP_TRY(hipEventRecord(g->chunk_done[i], is));
        if (is != cs) HIP_TRY(hipStreamWaitEvent(cs, g->chunk_done[i], 0));
        if (i == last_key_piece && J.try_keys) {
            // every key column is on the device (and converted): the key kernels of the whole block, once
            // (key columns are converted only when they are extended points; else their arrival is all the key kernels wait
            // for -- chunk_done[j] lies behind whatever else its compute stream was given before, i.e. behind unrelated hashes)
            for (size_t j = 0; j <= i; ++j)
                if (b.pieces[j].cols & COLS_KEYS) HIP_TRY(hipStreamWaitEvent(sl->key_stream, J.C.ext ? g->chunk_done[j] : g->chunk_up[j], 0));
            if (int rc = job_keys(J)) return rc;
        }
        if (!waiting.empty() && !hashes_blocked(i)) {
            for (const deferred& d : waiting) {
                if (needs_late) HIP_TRY(hipStreamWaitEvent(d.cs, g->chunk_up[last_late_piece], 0));
                if (keys_trail) HIP_TRY(hipStreamWaitEvent(d.cs, g->chunk_done[last_key_piece], 0));
                if (int rc = job_hash(J, d.first, d.count, d.cs)) return rc;
            }
            waiting.clear();
        }
        if (pc.cols & COLS_REST) {
            // the items of this piece are complete; their key columns may have been converted on the other stream
            for (size_t j = 0; j < i; ++j) {
                const host_piece& o = b.pieces[j];
                if ((o.cols & COLS_KEYS) && converted_on[j] != cs && o.first < pc.first + pc.count && pc.first < o.first + o.count)
                    HIP_TRY(hipStreamWaitEvent(cs, g->chunk_done[j], 0));
            }
            if (hashes_blocked(i)) waiting.push_back(deferred{pc.first, pc.count, cs});
            else if (int rc = job_hash(J, pc.first, pc.count, cs)) return rc;
        }
    }
    if (!waiting.empty()) return fail(JJS_ERR_ARG, "internal: ranges left waiting for the key kernels");
    HIP_TRY(hipStreamWaitEvent(g->stream, g->chunk_up[np - 1], 0));      // the equations read every column
    if (int rc = job_finish(J)) return rc;
#if defined(JJS_PROFILING)
    const double t_queued = now();
#endif
    if (status) HIP_TRY(hipMemcpyAsync(pst, st, nl, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipMemcpyAsync(ptally, g->tally, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipEventRecord(g->last_use, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
#if defined(JJS_PROFILING)
    const double t_drained = now();
#endif
    if (status) memcpy(status + b.lo, pst, nl);
    for (int k = 0; k < 4; ++k) b.tally[k] = ptally[k];
#if defined(JJS_PROFILING)
    g_host_timing[0] = t_stage; g_host_timing[1] = t_wait; g_host_timing[2] = now() - t_begin; g_host_timing[3] = (double)np;
    g_host_timing[4] = t_first; g_host_timing[5] = t_queued - t_begin; g_host_timing[6] = t_drained - t_queued; g_host_timing[7] = now() - t_drained;
#endif
    return JJS_OK;
}

// jjs_reserve: the device arena and the pinned staging a block of nl items with these columns needs (current device)
int reserve_host_block(const host_col* cols, size_t n_cols, size_t nl, int wire_points) {
    size_t row_keys = 0, row_rest = 0, row_late = 0, bytes = 0, largest = 256;
    for (size_t k = 0; k < n_cols; ++k) {
        (cols[k].group == COLS_KEYS ? row_keys : cols[k].group == COLS_LATE ? row_late : row_rest) += cols[k].width;
        bytes += pad256(nl * cols[k].width);
    }
    std::vector<host_piece> plan;
    plan_pieces(plan, largest, nl, row_keys, row_rest, row_late, wire_points, false);
    plan_pieces(plan, largest, nl, row_keys, row_rest, row_late, wire_points, true);
    if (int rc = ensure_stage(bytes + pad256(nl))) return rc;
    return ensure_pinned(HOST_SLOTS * pad256(largest) + pad256(nl) + 256);
}

// `unlocked`: the caller does NOT hold the engine's mutex (one driven device: the call holds that device's host_mu instead, so
// that calls of other threads -- resident ones, and host-buffer calls on the lanes -- are queued while this one uploads and waits)
int run_host(const host_col* cols, size_t n_cols, size_t n, uint8_t* status, uint64_t tally[4], call_builder build, int wire_points,
             bool unlocked = false) {
    if (n_cols > 8) return fail(JJS_ERR_ARG, "internal: too many columns");
    for (size_t k = 0; k < n_cols; ++k)
        if (n && !cols[k].p) return fail(JJS_ERR_ARG, "null input pointer");
    std::vector<device_state*> targets;
    if (L.devs.size() == 1) targets.push_back(g); else targets = L.devs;
    const size_t nd = targets.size();
    std::vector<host_block> blocks(nd);
    device_restore restore;
    const size_t per = (n + nd - 1) / nd;
    // staging helpers: the host cores this process may use, shared among the devices it drives
    unsigned staging_threads = 1;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        const unsigned cores = sched_getaffinity(0, sizeof(set), &set) == 0 ? (unsigned)CPU_COUNT(&set) : 1u;
        staging_threads = cores / (unsigned)nd;
        if (staging_threads > HOST_STAGING_THREADS_MAX) staging_threads = HOST_STAGING_THREADS_MAX;
        if (staging_threads < 1) staging_threads = 1;
    }
    size_t row_keys = 0, row_rest = 0, row_late = 0;
    for (size_t k = 0; k < n_cols; ++k)
        (cols[k].group == COLS_KEYS ? row_keys : cols[k].group == COLS_LATE ? row_late : row_rest) += cols[k].width;
    for (size_t d = 0; d < nd; ++d) {
        host_block& b = blocks[d];
        b.lo = d * per < n ? d * per : n;
        b.hi = b.lo + per < n ? b.lo + per : n;
        b.largest_bytes = 256;
        plan_pieces(b.plans[0], b.largest_bytes, b.hi - b.lo, row_keys, row_rest, row_late, wire_points, false);
        plan_pieces(b.plans[1], b.largest_bytes, b.hi - b.lo, row_keys, row_rest, row_late, wire_points, true);
        b.staging_threads = staging_threads;
        b.unlocked = unlocked && nd == 1;
    }
    auto work = [&](size_t d) {
        host_block& b = blocks[d];
        verify_job J;
        b.rc = no_throw([&] { return run_host_block(targets[d], cols, n_cols, b, status, build, J); });
        if (b.rc != JJS_OK) {
            // leave nothing in flight into the caller's arrays, the pinned slots or the counters
            snprintf(b.err, sizeof(b.err), "%s", t_err);
            job_abandon(J);
            (void)hipStreamSynchronize(targets[d]->stream);
            for (hipStream_t side : targets[d]->side) (void)hipStreamSynchronize(side);
            for (hipStream_t is : targets[d]->ingest) (void)hipStreamSynchronize(is);
            (void)hipStreamSynchronize(targets[d]->copy_stream);
        }
        if (b.owned) {
            std::lock_guard<std::mutex> lock(L.mu);
            b.owned->host_owned = false;
            b.owned = nullptr;
        }
    };
    if (nd == 1) {
        work(0);
    } else {
        // one thread per device; a thread that cannot be started is not fatal: its block runs on this thread afterwards
        std::vector<std::thread> threads;
        std::vector<size_t> here;
        for (size_t d = 0; d < nd; ++d) {
            try { threads.emplace_back(work, d); } catch (...) { here.push_back(d); }
        }
        for (size_t d : here) work(d);
        for (std::thread& t : threads) t.join();
    }
    g = targets[0];
    for (size_t d = 0; d < nd; ++d)
        if (blocks[d].rc != JJS_OK) return fail(blocks[d].rc, "device %d: %s", targets[d]->device, blocks[d].err);
    if (nd > 1 && L.comms_up)
        if (int rc = allreduce_tallies()) {
            for (size_t d = 0; d < nd; ++d) { (void)hipSetDevice(targets[d]->device); (void)hipStreamSynchronize(targets[d]->stream); }
            return rc;
        }
    if (nd > 1 && L.comms_up) {                // every device now holds the sum: fetch it again
        for (size_t d = 0; d < nd; ++d) {
            HIP_TRY(hipSetDevice(targets[d]->device));
            HIP_TRY(hipMemcpyAsync(blocks[d].tally, targets[d]->tally, sizeof(blocks[d].tally), hipMemcpyDeviceToHost, targets[d]->stream));
            HIP_TRY(hipEventRecord(targets[d]->last_use, targets[d]->stream));
        }
        int rc = JJS_OK;
        for (size_t d = 0; d < nd; ++d) {          // drain every device even if one of them reports an error
            hipError_t e = hipSetDevice(targets[d]->device);
            if (e == hipSuccess) e = hipStreamSynchronize(targets[d]->stream);
            if (e != hipSuccess && rc == JJS_OK) rc = fail(JJS_ERR_HIP, "device %d: %s", targets[d]->device, hipGetErrorString(e));
        }
        if (rc != JJS_OK) return rc;
    }
    if (tally) {
        for (int i = 0; i < 4; ++i) tally[i] = blocks[0].tally[i];
        // test mode (logical devices sharing one GPU cannot form an RCCL clique): add the counters here
        if (nd > 1 && !L.comms_up)
            for (size_t d = 1; d < nd; ++d)
                for (int i = 0; i < 4; ++i) tally[i] += blocks[d].tally[i];
    }
    return JJS_OK;
}
