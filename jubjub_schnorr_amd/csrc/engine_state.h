// Part of jjs_gpu.hip (included inside its anonymous namespace): what the library owns between jjs_init and jjs_shutdown --
// per-device state, call slots, the staging threads of the host-buffer calls, the RCCL handle -- and the helpers every
// launch uses (errors, grids, slot ordering, grow-only buffers, the latency path's launch).
#pragma once
// ---------------------------------------------------------------------------------------------
// Per-device state: everything a launch on that device needs (tables, per-lane workspace, scratch).
// What one verification call in flight needs beside its inputs.  A device has one big slot (slot 0: the workspace
// of the persistent verify grid, used by every large call and by the signer / multisig kernels), N_SMALL_SLOTS
// small ones, handed out round-robin to calls of at most SMALL_SLOT_ITEMS items, and N_MEDIUM_SLOTS medium ones for
// calls of at most MEDIUM_SLOT_ITEMS: calls in different slots touch disjoint buffers and are not ordered against
// each other, so small and medium calls issued on different streams overlap on the device; calls that share a slot
// are ordered by its event.
// Helper threads of the host-buffer entry points (pageable -> pinned staging copies; one thread moves ~11 GB/s).  They are
// started once per device and parked on a condition variable between pieces; a thread that cannot be created is simply
// missing (the caller takes its share), nothing here throws past the extern "C" boundary.
class staging_pool {
    std::mutex mu;
    std::condition_variable work_cv, done_cv;
    std::vector<std::thread> threads;
    void (*fn)(void*, unsigned) = nullptr;
    void* ctx = nullptr;
    unsigned tasks = 0, next = 0, running = 0;
    uint64_t epoch = 0;
    bool quit = false;
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        uint64_t seen = 0;
        for (;;) {
            work_cv.wait(lk, [&] { return quit || (epoch != seen && next < tasks); });
            if (quit) return;
            seen = epoch;
            while (next < tasks) {
                const unsigned t = next++;
                ++running;
                lk.unlock();
                fn(ctx, t);
                lk.lock();
                --running;
            }
            if (running == 0) done_cv.notify_all();
        }
    }
public:
    explicit staging_pool(unsigned helpers) {
        for (unsigned i = 0; i < helpers; ++i) {
            try { threads.emplace_back([this] { loop(); }); } catch (...) { break; }
        }
    }
    ~staging_pool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        work_cv.notify_all();
        for (std::thread& t : threads) t.join();
    }
    unsigned helpers() const { return (unsigned)threads.size(); }
    // f(c, t) for t = 0 .. T-1: begin() hands the tasks to the helpers and returns; join() lets the caller take what is
    // left and returns when every task is done.  One batch of tasks at a time.
    void begin(unsigned T, void (*f)(void*, unsigned), void* c) {
        std::lock_guard<std::mutex> lk(mu);
        fn = f; ctx = c; tasks = T; next = 0; ++epoch;
        work_cv.notify_all();
    }
    void join() {
        std::unique_lock<std::mutex> lk(mu);
        while (next < tasks) {
            const unsigned t = next++;
            ++running;
            lk.unlock();
            fn(ctx, t);
            lk.lock();
            --running;
        }
        done_cv.wait(lk, [&] { return running == 0; });
        tasks = 0; fn = nullptr; ctx = nullptr;
    }
};

// What the last key-table attempt of a slot found (key_params::counters), copied to pinned host memory behind the call:
// the host reads it before the slot's next attempt (note_key_feedback) -- no call ever waits for it.
struct key_feedback {
    uint32_t counters[8];
};
struct call_slot {
    uint32_t* workspace = nullptr;    // WS_WORDS_PER_LANE words per lane of the verify grid
    int grid_verify = 0;              // blocks of verify_kernel that fit this workspace
    uint64_t* pending = nullptr;      // queue of the resolve pass: [0] = count, then one entry per queued item
    size_t pending_items = 0;
    uint8_t* prep = nullptr;          // prepare_kernel -> verify_kernel records, 65 bytes per item (grow-only)
    size_t prep_items = 0;
    uint8_t* wire = nullptr;          // decoded / normalised points (4 x n x 64), flags, scratch: wire and ext entry points
    size_t wire_items = 0;
    uint8_t* small = nullptr;         // latency path: window tables of the chain lanes + per-point verdicts (grow-only)
    size_t small_bytes = 0;
    // key-table path (big and medium slots).  Two arenas, both grow-only: the index (hash tables, key ids, item order:
    // sized by the batch) and the pool of per-key bases and window tables, which is sized by the number of distinct keys
    // the slot's calls have carried -- KEY_POOL_INITIAL_BYTES to begin with, more once a call has shown that it needs more.
    uint8_t* keys = nullptr;
    size_t keys_bytes = 0;
    uint8_t* key_pool = nullptr;
    size_t key_pool_bytes = 0;
    size_t key_pool_want = 0;         // what the last call that found the pool too small would have needed
    size_t key_pool_refused = 0;      // a size hipMalloc turned down (not asked for again)
    key_feedback* seen = nullptr;     // pinned host memory
    bool seen_pending = false;        // `seen` is being written by a call that may still run (its end: last_use)
    bool keys_repeated = true;        // the slot's last key-table attempt that has ended built tables (launch_staged)
    uint64_t seen_n = 0;              // ... whose batch had this many items in
    uint32_t seen_cols = 0;           // ... this many key columns
    hipStream_t key_stream = nullptr; // the per-key kernels of the slot's call run here, beside the challenge hashes
    hipEvent_t key_fork = nullptr, key_mid = nullptr, key_join = nullptr, key_ahead = nullptr, key_chains = nullptr;
    hipStream_t table_stream = nullptr;   // key_table_kernel runs here, at the lowest priority: see job_keys
    hipEvent_t last_use = nullptr;    // end of the last launch that used this slot
    hipStream_t last_stream = nullptr;// ... and the stream it was issued on
};
constexpr int N_SMALL_SLOTS = 3;
constexpr size_t SMALL_SLOT_ITEMS = 16384;
// Calls of up to MEDIUM_SLOT_ITEMS items take one of N_MEDIUM_SLOTS medium slots in turn: such a call is a few waves
// per SIMD at most and is bound by the latency of one signature (~1.7 ms), so calls on different streams overlap almost
// freely.  A medium slot has everything the big one has (workspace, key arena) for its size.
constexpr int N_MEDIUM_SLOTS = 3;
constexpr size_t MEDIUM_SLOT_ITEMS = 131072;
// Larger calls take two big slots in turn: what two big batches in flight gain is each other's latency-bound stretches
// (key dedup, the per-key doubling chains, the resolve pass), ~10 % of a batch, filled with the other's arithmetic.
constexpr int N_BIG_SLOTS = 2;
constexpr int SECOND_BIG_SLOT = 1 + N_SMALL_SLOTS + N_MEDIUM_SLOTS;
constexpr int N_SLOTS = 1 + N_SMALL_SLOTS + N_MEDIUM_SLOTS + (N_BIG_SLOTS - 1);
// largest batch the latency path takes, by number of equations (1: single, 2: double).  One call of 32 768 items would
// still return sooner on this path (1.27 against 1.56 ms single; tools/batch_size_curve.py), but it does twice the work:
// callers who keep several such calls in flight get 38 M/s from the throughput path and 28 M/s from this one
// (tools/concurrent_calls.py), so the limit stays where the chip is not yet full
constexpr size_t SMALL_PATH_MAX_ITEMS[3] = {0, 16384, 16384};
// up to here the scalars are cut into 8 pieces instead of 4 (small_batch.h): shorter tail, twice the chain work
constexpr size_t SMALL_PATH_FINE_ITEMS[3] = {0, 4096, 4096};
// the per-item-generator scheme (full-size scalars on two variable points: the chains are twice as long)
constexpr size_t SMALL_PATH_MAX_ITEMS_VARGEN = 16384, SMALL_PATH_FINE_ITEMS_VARGEN = 4096;

constexpr size_t HOST_MAX_PIECES = 64;      // pieces a host-buffer call uploads its block in (plan_pieces)
#ifndef JJS_HOST_SIDE_STREAMS
#define JJS_HOST_SIDE_STREAMS 2
#endif
constexpr int HOST_SIDE_STREAMS = JJS_HOST_SIDE_STREAMS;
struct device_state {
    int device = -1;               // HIP device ordinal
    call_slot slots[N_SLOTS];          // [0] big, then the small ones, then the medium ones, then the second big one
    unsigned next_small = 0, next_medium = 0, next_big = 0;
    hipStream_t stream = nullptr;  // used by the host-buffer entry points
    hipStream_t side[HOST_SIDE_STREAMS] = {};   // ... whose ranges go to `stream` and these in turn (run_host_block)
    hipStream_t ingest[2] = {};                 // ... and whose extended points are normalised here, ahead of the hashes (priority)
    hipEvent_t host_begin = nullptr;
    uint32_t* comb_g = nullptr;
    uint32_t* comb_gn = nullptr;
    uint8_t* tag = nullptr;
    unsigned long long* tally = nullptr;
    int grid_sign = 0, grid_resolve = 0, grid_prepare = 0, grid_key_verify = 0;
    hipEvent_t last_use = nullptr;  // host-buffer calls: end of the last use of the staging arena and the counters
    uint32_t* dlog_pow = nullptr;  // square-root tables (decode.h)
    uint8_t* dlog_hash = nullptr;
    uint32_t* tags_long = nullptr; // SAFE tags for long transcripts (multisig)
    uint8_t* msig = nullptr;       // multisig scratch
    size_t msig_items = 0, msig_transcripts = 0;
    int grid_msig = 0;
    int key_priority = 0;                // stream priority of the slots' key streams
    int table_priority = 0;              // ... and of their table streams (the lowest)
    uint64_t stats[JJS_PATH_STATS] = {}; // jjs_path_stats: which path the calls on this device took
    hipStream_t copy_stream = nullptr;   // host-buffer calls: uploads and status downloads, beside `stream`
    hipEvent_t side_join = nullptr, ingest_done = nullptr;
    staging_pool* stagers = nullptr;     // host-buffer calls: the threads that copy pageable -> pinned with the caller's
    uint8_t* stage = nullptr;            // host-buffer calls: device copies of the inputs + statuses (grow-only)
    size_t stage_bytes = 0;
    uint8_t* pinned = nullptr;           // host-buffer calls: pinned host staging (two input slots + statuses, grow-only)
    size_t pinned_bytes = 0;
    hipEvent_t chunk_up[HOST_MAX_PIECES] = {}, chunk_done[HOST_MAX_PIECES] = {};   // per piece of a host-buffer call: uploaded, converted
};

// RCCL is needed only when one process drives several devices, so it is loaded on demand.
struct rccl_api {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

constexpr int MAX_DEVICES = 16;

struct library_state {
    std::mutex mu;
    std::vector<device_state*> devs;       // devices this process drives (jjs_init)
    bool virtual_devices = false;          // test mode: several logical devices on one physical device
    rccl_api rccl;
    ncclComm_t comms[MAX_DEVICES] = {};
    bool comms_up = false;
};
library_state L;
// Device bound to the work in progress on THIS host thread: set by check_ready for an entry point and by each
// per-device worker of run_host for its own block (the workers run concurrently, one device each).
thread_local device_state* g = nullptr;
thread_local call_slot* sl = nullptr;      // slot of the call in progress on this thread (pick_slot)

// One message buffer per host thread: jjs_last_error() describes the calling thread's last failure and a
// pointer it returned is never written by another thread.
thread_local char t_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(x)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) return fail(JJS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); \
    } while (0)
#define RCCL_TRY(x)                                                                                      \
    do {                                                                                                 \
        ncclResult_t r_ = (x);                                                                           \
        if (r_ != ncclSuccess) return fail(JJS_ERR_COLLECTIVE, "%s: %s", #x, L.rccl.GetErrorString(r_)); \
    } while (0)

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int grid_for(int resident, size_t n) {
    size_t want = (n + BLOCK - 1) / BLOCK;
    if (want < 1) want = 1;
    return (int)(want < (size_t)resident ? want : (size_t)resident);
}

#if defined(JJS_PROFILING)
uint32_t g_skip_phases = 0;       // set by jjs_debug_skip_phases (libjjs_gpu_prof.so only)
bool g_allow_virtual = false;     // set by jjs_debug_allow_virtual_devices (libjjs_gpu_prof.so only)
int g_force_path = 0;             // set by jjs_debug_force_path: 0 = by size, 1 = throughput path, 2 = latency path
int g_force_positions = 0;        // ... and 4 or 8 pieces on the latency path (0 = by size)
double g_host_timing[8] = {};     // last host-buffer call, seconds: see jjs_debug_host_timing (include/jjs_gpu_profiling.h)
bool g_keep_order = false;        // ... 0x1000: key-table path without grouping the items by key
int g_force_window = 0;           // ... 5: narrow windows on the key-table path whatever the signatures per key
bool g_fail_key_arena = false;    // set by jjs_debug_fail_key_arena: the key-table pool "cannot be allocated"
bool g_pin_hash_seed = false;     // set by jjs_debug_pin_hash_seed: the dedup hash runs with seed 0
#endif

// Small calls take the small slots in turn, everything else the big one (see call_slot).
void pick_slot(size_t n, hipStream_t s) {
    if (n <= SMALL_SLOT_ITEMS) { sl = &g->slots[1 + g->next_small]; g->next_small = (g->next_small + 1) % N_SMALL_SLOTS; }
    else if (n <= MEDIUM_SLOT_ITEMS) { sl = &g->slots[1 + N_SMALL_SLOTS + g->next_medium]; g->next_medium = (g->next_medium + 1) % N_MEDIUM_SLOTS; }
    else {
        // a big slot is a big arena: calls that follow each other on one stream are ordered anyway and stay in one
        // slot; a call from another stream takes the other one if this one is still busy
        call_slot &a = g->slots[0], &b = g->slots[SECOND_BIG_SLOT];
        if (a.last_stream == s) sl = &a;
        else if (b.last_stream == s) sl = &b;
        else sl = hipEventQuery(a.last_use) == hipSuccess ? &a : (hipEventQuery(b.last_use) == hipSuccess ? &b : (g->next_big++ % N_BIG_SLOTS ? &b : &a));
    }
    sl->last_stream = s;
}
void big_slot() { sl = &g->slots[0]; sl->last_stream = nullptr; }
// Launches that use one slot are ordered one after the other on the device, also across streams: each waits
// for the slot's previous user.
int begin_shared(hipStream_t s) {
    HIP_TRY(hipStreamWaitEvent(s, sl->last_use, 0));
    return JJS_OK;
}
int end_shared(hipStream_t s) {
    HIP_TRY(hipEventRecord(sl->last_use, s));
    return JJS_OK;
}

int ensure_pending(size_t n) {
    if (n <= sl->pending_items) return JJS_OK;
    if (sl->pending) {
        HIP_TRY(hipDeviceSynchronize());        // earlier launches may still use the old queue
        HIP_TRY(hipFree(sl->pending));
        sl->pending = nullptr; sl->pending_items = 0;
    }
    size_t cap = n < SMALL_SLOT_ITEMS ? SMALL_SLOT_ITEMS : n;
    HIP_TRY(hipMalloc(&sl->pending, (cap + 2) * sizeof(uint64_t)));
    sl->pending_items = cap;
    return JJS_OK;
}

int ensure_prep(size_t n) {
    if (n <= sl->prep_items) return JJS_OK;
    if (sl->prep) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(sl->prep));
        sl->prep = nullptr; sl->prep_items = 0;
    }
    size_t cap = n < SMALL_SLOT_ITEMS ? SMALL_SLOT_ITEMS : n;
    HIP_TRY(hipMalloc(&sl->prep, cap * 65 + 64));
    sl->prep_items = cap;
    return JJS_OK;
}

int ensure_small(size_t bytes) {
    if (bytes <= sl->small_bytes) return JJS_OK;
    if (sl->small) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(sl->small));
        sl->small = nullptr; sl->small_bytes = 0;
    }
    HIP_TRY(hipMalloc(&sl->small, bytes));
    sl->small_bytes = bytes;
    return JJS_OK;
}

// Latency path (small_batch.h): two launches, every signature spread over 11 (single) or 21 (double) lanes.
int launch_small(verify_params P, hipStream_t s) {
    const bool vargen = P.eq[0].comb == nullptr;
    uint32_t positions = P.n <= (vargen ? SMALL_PATH_FINE_ITEMS_VARGEN : SMALL_PATH_FINE_ITEMS[P.n_eq]) ? 8 : 4;
#if defined(JJS_PROFILING)
    if (g_force_positions) positions = (uint32_t)g_force_positions;
#endif
    const size_t table_bytes = P.n * sb_table_words_per_item(P.n_eq, positions) * sizeof(uint32_t);
    if (int rc = ensure_small(table_bytes + 4 * P.n + 64)) return rc;
    small_params S{};
    P.small_mode = 1;
    S.V = P;
    S.tables = reinterpret_cast<uint32_t*>(sl->small);
    S.point_ok = sl->small + table_bytes;
    S.positions = positions;
    S.windows = vargen ? 64 : 32;
    // eight lanes per hash where the hash is the critical path (fixed generator) and the batch leaves lanes idle
    S.hash_lanes = (!vargen && P.n <= SMALL_PATH_FINE_ITEMS[P.n_eq]) ? SB_HASH_LANES : 1;
    const unsigned hash_blocks = (unsigned)((P.n * S.hash_lanes + BLOCK - 1) / BLOCK);
    const unsigned chain_blocks = (unsigned)((P.n * P.n_eq * 2 + BLOCK - 1) / BLOCK);
    const unsigned point_blocks = (unsigned)((P.n * P.n_points + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(small_a_kernel, dim3(hash_blocks + positions * chain_blocks + point_blocks), dim3(BLOCK), 0, s, S,
                       (uint32_t)hash_blocks, (uint32_t)chain_blocks);
    hipLaunchKernelGGL(small_b_kernel, dim3((unsigned)((P.n * P.n_eq * positions + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, S);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
