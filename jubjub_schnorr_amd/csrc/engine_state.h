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
    hipEvent_t key_fork = nullptr, key_mid = nullptr, key_join = nullptr, key_ahead = nullptr, key_chains = nullptr, key_cleared = nullptr;
    hipStream_t table_stream = nullptr;   // key_table_kernel runs here, at the lowest priority: see job_keys
    hipEvent_t last_use = nullptr;    // end of the last launch that used this slot
    hipStream_t last_stream = nullptr;// ... and the stream it was issued on
    bool host_owned = false;          // a large host-buffer call is feeding this slot right now, outside the engine's mutex (run_host)
};
#ifndef JJS_SMALL_SLOTS
#define JJS_SMALL_SLOTS 6
#endif
constexpr int N_SMALL_SLOTS = JJS_SMALL_SLOTS;   // six since round 4: host threads with a few signatures per call keep more than three in flight
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
// up to here eight lanes share the challenge hash of a fixed-generator signature (launch_small)
constexpr size_t SMALL_PATH_COOP_ITEMS = 6144;
// the per-item-generator scheme (full-size scalars on two variable points: the chains are twice as long)
constexpr size_t SMALL_PATH_MAX_ITEMS_VARGEN = 16384, SMALL_PATH_FINE_ITEMS_VARGEN = 4096;

// Host-buffer calls of at most LANE_MAX_ITEMS items do not go through the piece-by-piece pipeline of the large ones
// (host_calls.h): each runs on a LANE -- a stream, a pinned host area and a device area -- and holds the engine's mutex only
// while it joins the lane and while its kernels are queued (some tens of microseconds), not while it copies or waits.  So T
// service threads that each verify a few signatures per call -- what the reference's callers do (one item per call,
// src/keys/public.rs:114-118) -- overlap on the device like T streams do.
// And they COMBINE: calls of at most COMBINE_MAX_CALL_ITEMS items of one scheme and input format that arrive while a lane
// launch of that shape runs append their items to the lane that is filling, and that lane goes to the device as ONE launch
// (of up to COMBINE_CAP_ITEMS items) when the launch ahead of it has ended -- a launch of 4 096 signatures returns as soon as
// one of 1 024: the latency path has lanes to spare at these sizes, so eight threads are served in the time of one.  A call
// that finds no launch of its shape running is launched at once, alone: one thread's latency is what it was.  Launches of
// different shapes, and calls too large to combine, run side by side.  Every caller gets its own statuses and its own tally
// (counted on the host from its statuses).
struct host_lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;        // end of the launch in progress on the lane
    uint8_t* dev = nullptr;           // device area: the columns of the launch, its statuses (grow-only)
    size_t dev_bytes = 0;
    uint8_t* pinned = nullptr;        // pinned host area, same layout (grow-only)
    size_t pinned_bytes = 0;
    // the launch that is being put together / runs on the lane
    enum : int { FREE = 0, OPEN = 1, LAUNCHED = 2, DONE = 3 };
    int state = FREE;
    int scheme = -1, format = -1;     // call shape of the members
    bool combinable = false;          // further calls may join while it is OPEN
    bool combinable_shape = false;    // a launch of combinable calls (LAUNCHED: no second one of its shape beside it)
    size_t cap = 0, items = 0;        // items the layout holds / items the members have claimed
    unsigned copying = 0;             // members still copying their rows in (under the engine's mutex)
    std::atomic<unsigned> members{0}; // written under the mutex; the leader reads it while it holds the window open
    std::chrono::steady_clock::time_point gather_until{};    // not launched before (see COMBINE_WINDOW_US) ...
#if defined(JJS_LANE_TRACE)
    std::chrono::steady_clock::time_point trace_open{}, trace_done{};
#endif
    unsigned expect = 0;              // ... unless it has this many members: everybody the window is held open for has arrived
    int rc = 0;                       // outcome of the launch (JJS_OK or the error every member returns)
    char err[256] = "";
    // bumped when the launch has ended (state DONE): what the members other than the lane's leader wait for, without the
    // engine's mutex (host_lanes.h lane_await_done)
    std::atomic<uint32_t> done_gen{0};
};
constexpr int N_HOST_LANES = 8;
// A lane has ONE leader, the call that opened it: it waits for the lane to be complete and for its turn, launches, waits for
// the device and publishes the outcome.  The other members only wait for `done_gen` -- polling it, as long as fewer than
// LANE_MAX_SPINNERS threads of the process are polling for a lane already, else asleep on it (futex): a service with more
// threads than cores (64 threads with one signature each is what the reference's one-item API makes of a busy node) must not
// spend its cores on waiting, nor wake every waiter for every change of every lane.
#ifndef JJS_LANE_MAX_SPINNERS
#define JJS_LANE_MAX_SPINNERS 8
#endif
constexpr int LANE_MAX_SPINNERS = JJS_LANE_MAX_SPINNERS;
// The lanes take the calls of the latency path (at most 16 384 items: a copy of a few megabytes by the calling thread).  A
// larger call goes through the piece-by-piece pipeline of host_calls.h, whose staging threads and overlapped uploads it needs
// (131 072 single signatures: 2.45 ms there, 3.3-3.7 ms on a lane with one thread copying 25 MB ahead of one upload).
#ifndef JJS_LANE_MAX_ITEMS
#define JJS_LANE_MAX_ITEMS 16384
#endif
constexpr size_t LANE_MAX_ITEMS = JJS_LANE_MAX_ITEMS;
#ifndef JJS_COMBINE_MAX_CALL_ITEMS
#define JJS_COMBINE_MAX_CALL_ITEMS 4096
#endif
constexpr size_t COMBINE_MAX_CALL_ITEMS = JJS_COMBINE_MAX_CALL_ITEMS, COMBINE_CAP_ITEMS = 16384;      // the cap: what the latency path takes
// When a launch ends, the lane that has been filling behind it is not launched before this many microseconds have passed:
// the threads that launch served come back within microseconds of each other and find it still open.  (Launched at once, it
// would leave without them, and the threads would take turns in half-empty launches.)  The same holds for a lane that is
// opened just behind a launch that carried several calls (its first returning caller): two threads would otherwise alternate
// between launches of two calls and of one.  The window closes early when as many calls have joined as the launch released
// (plus those that were waiting already).  A caller that finds no launch of its shape running, and none of several calls
// just ended, does not wait.
#ifndef JJS_COMBINE_WINDOW_US
#define JJS_COMBINE_WINDOW_US 50
#endif
constexpr unsigned COMBINE_WINDOW_US = JJS_COMBINE_WINDOW_US;
#ifndef JJS_COMBINE_EXPECT
#define JJS_COMBINE_EXPECT 1        // 0 (A/B builds): the window is always held for its whole length, and only behind a running launch
#endif
// A buffer that was replaced by a larger one.  It may still be in use by launches that are in flight, and hipFree would
// wait for every stream of the device: it is kept until jjs_trim / jjs_shutdown (grow-only buffers grow geometrically, so
// what is kept is less than what is live).
struct retired_buffer { void* p; bool host; size_t bytes; };

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
    std::atomic<uint64_t> stats[JJS_PATH_STATS] = {};   // jjs_path_stats: which path the calls on this device took
    std::mutex host_mu;                  // large host-buffer calls: one at a time per device (they share the staging below)
    std::mutex retired_mu;               // guards `retired`
    hipStream_t copy_stream = nullptr;   // host-buffer calls: uploads and status downloads, beside `stream`
    hipEvent_t side_join = nullptr, ingest_done = nullptr;
    staging_pool* stagers = nullptr;     // host-buffer calls: the threads that copy pageable -> pinned with the caller's
    uint8_t* stage = nullptr;            // host-buffer calls: device copies of the inputs + statuses (grow-only)
    size_t stage_bytes = 0;
    uint8_t* pinned = nullptr;           // host-buffer calls: pinned host staging (two input slots + statuses, grow-only)
    size_t pinned_bytes = 0;
    hipEvent_t chunk_up[HOST_MAX_PIECES] = {}, chunk_done[HOST_MAX_PIECES] = {};   // per piece of a host-buffer call: uploaded, converted
    host_lane lanes[N_HOST_LANES];       // small and medium host-buffer calls: one lane per launch in flight or filling
    std::atomic<uint64_t> lane_epoch{0}; // bumped (under the engine's mutex) whenever a lane changes state: what lane leaders and callers without a lane poll
    std::atomic<int> lane_spinners{0};   // threads polling for a lane right now (at most LANE_MAX_SPINNERS; the others sleep)
    size_t lane_last_items[3][3] = {};   // [scheme][format]: items of the last combined launch (sizes the next lane)
    unsigned lane_last_members[3][3] = {};   // ... the calls it carried, and when it ended: the callers it released are about to
    std::chrono::steady_clock::time_point lane_last_done[3][3] = {};   // come back, and the next lane waits for them (COMBINE_WINDOW_US)
    std::vector<retired_buffer> retired; // replaced buffers, freed by jjs_trim / jjs_shutdown
    size_t retired_bytes = 0;
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
    std::condition_variable lane_cv;       // a host lane has changed state (waited for under `mu`)
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
thread_local call_slot* forced_slot = nullptr;   // jjs_reserve: the slot the next builder sizes, instead of the one whose turn it is

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

// Multisignature passes that hash (delinearisation, a, c) take eight lanes per item while that still leaves the device a wave per
// SIMD: a transcript's sponge is ONE chain of (3 + 4 n) / 4 permutations, and eight lanes run a permutation in 0.12 ms instead
// of 0.25 (hades29.h, coop).  Beyond, the passes are throughput-bound and eight lanes per item would be eight times the work.
constexpr size_t MSIG_COOP_MAX_ITEMS = 8192;
int grid_for(int resident, size_t n) {
    size_t want = (n + BLOCK - 1) / BLOCK;
    if (want < 1) want = 1;
    return (int)(want < (size_t)resident ? want : (size_t)resident);
}

#if defined(JJS_PROFILING)
uint32_t g_skip_phases = 0;       // set by jjs_debug_skip_phases (libjjs_gpu_prof.so only)
bool g_allow_virtual = false;     // set by jjs_debug_allow_virtual_devices (libjjs_gpu_prof.so only)
int g_force_path = 0;             // set by jjs_debug_force_path: 0 = by size, 1 = throughput path, 2 = latency path
int g_force_positions = 0;        // ... and 4, 8 or 16 pieces on the latency path (0 = by size)
double g_host_timing[8] = {};     // last host-buffer call, seconds: see jjs_debug_host_timing (include/jjs_gpu_profiling.h)
bool g_keep_order = false;        // ... 0x1000: key-table path without grouping the items by key
int g_force_window = 0;           // ... 5: narrow windows on the key-table path whatever the signatures per key
bool g_fail_key_arena = false;    // set by jjs_debug_fail_key_arena: the key-table pool "cannot be allocated"
bool g_pin_hash_seed = false;     // set by jjs_debug_pin_hash_seed: the dedup hash runs with seed 0
#endif

// Small calls take the small slots in turn, everything else the big one (see call_slot).
void pick_slot(size_t n, hipStream_t s) {
    if (forced_slot) { sl = forced_slot; return; }
    if (n <= SMALL_SLOT_ITEMS) { sl = &g->slots[1 + g->next_small]; g->next_small = (g->next_small + 1) % N_SMALL_SLOTS; }
    else if (n <= MEDIUM_SLOT_ITEMS) {
        // (not a slot that a host-buffer call is feeding outside the engine's mutex: at most one is)
        for (int turn = 0; turn < N_MEDIUM_SLOTS; ++turn) {
            sl = &g->slots[1 + N_SMALL_SLOTS + g->next_medium];
            g->next_medium = (g->next_medium + 1) % N_MEDIUM_SLOTS;
            if (!sl->host_owned) break;
        }
    }
    else {
        // a big slot is a big arena: calls that follow each other on one stream are ordered anyway and stay in one
        // slot; a call from another stream takes the other one if this one is still busy
        call_slot &a = g->slots[0], &b = g->slots[SECOND_BIG_SLOT];
        // a large host-buffer call feeds its slot outside the engine's mutex: it always takes the second big slot (the first
        // also serves the signer, the decoder and the multisig kernels), and nobody else takes that one meanwhile
        if (s == g->stream) sl = &b;
        else if (b.host_owned) sl = &a;
        else if (a.last_stream == s) sl = &a;
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

// Grow-only buffers: the replacement is allocated first, the old buffer is retired (launches in flight may still use it; it
// is freed by jjs_trim / jjs_shutdown).  No call waits for the device here.
size_t grown(size_t want) {              // the smallest of 2^k, 1.5 * 2^k that holds `want`: what is retired stays below what is live
    size_t cap = 4096;
    while (cap < want) cap <<= 1;
    const size_t mid = cap / 4 * 3;
    return mid >= want ? mid : cap;
}
void retire(void* p, bool host, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g->retired_mu);
    try {
        g->retired.push_back(retired_buffer{p, host, bytes});
        g->retired_bytes += bytes;
    } catch (...) {                       // no room for the note: wait for the device and free it now
        (void)hipDeviceSynchronize();
        if (host) (void)hipHostFree(p); else (void)hipFree(p);
    }
}
void free_retired(device_state& d) {      // the caller has made sure that the device is idle
    std::lock_guard<std::mutex> lock(d.retired_mu);
    for (retired_buffer& r : d.retired) {
        if (r.host) (void)hipHostFree(r.p); else (void)hipFree(r.p);
    }
    d.retired.clear();
    d.retired_bytes = 0;
}
template <typename T>
int regrow(T*& buf, size_t& have, size_t old_bytes, size_t want_units, size_t bytes) {
    T* fresh = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fresh), bytes));
    retire(buf, false, old_bytes);
    buf = fresh;
    have = want_units;
    return JJS_OK;
}

int ensure_pending(size_t n) {
    if (n <= sl->pending_items) return JJS_OK;
    const size_t cap = grown(n < SMALL_SLOT_ITEMS ? SMALL_SLOT_ITEMS : n);
    return regrow(sl->pending, sl->pending_items, (sl->pending_items + 2) * sizeof(uint64_t), cap, (cap + 2) * sizeof(uint64_t));
}

int ensure_prep(size_t n) {
    if (n <= sl->prep_items) return JJS_OK;
    const size_t cap = grown(n < SMALL_SLOT_ITEMS ? SMALL_SLOT_ITEMS : n);
    return regrow(sl->prep, sl->prep_items, sl->prep_items * 65 + 64, cap, cap * 65 + 64);
}

int ensure_small(size_t bytes) {
    if (bytes <= sl->small_bytes) return JJS_OK;
    const size_t cap = grown(bytes);
    return regrow(sl->small, sl->small_bytes, sl->small_bytes, cap, cap);
}

// Latency path (small_batch.h): two launches, every signature spread over 11 (single) or 21 (double) lanes.
// A call that is alone on the device spreads every signature over as many lanes as shorten its critical path: 8 pieces per
// scalar and eight lanes per hash up to 4 096 items (small_batch.h).  That buys latency with work -- a call of 1 024 single
// signatures is 416 waves instead of 180 -- and the device holds about one such call per 256 compute units at full speed: with
// more in flight their waves share SIMDs and every call takes longer (four host threads of such calls: 1.9 x one thread's
// rate).  So a small call that finds `others` small calls of other slots still running when it is queued takes the
// economical cut (4 pieces, one lane per hash) from SMALL_ECONOMY_FROM others on.  Statuses do not depend on the cut.
#ifndef JJS_SMALL_ECONOMY_FROM
#define JJS_SMALL_ECONOMY_FROM 1
#endif
constexpr unsigned SMALL_ECONOMY_FROM = JJS_SMALL_ECONOMY_FROM;
constexpr size_t SMALL_QUAD_CHAIN_MAX_ITEMS = 2048, SMALL_QUAD_CHAIN_MAX_ITEMS_FIXED = 1024;      // see launch_small (per-item generator, fixed generator)
unsigned small_calls_in_flight() {
    unsigned k = 0;
    for (int i = 1; i <= N_SMALL_SLOTS; ++i) {
        call_slot& c = g->slots[i];
        if (&c != sl && hipEventQuery(c.last_use) == hipErrorNotReady) ++k;
    }
    (void)hipGetLastError();              // "not ready" is an answer, not a failure of this call
    return k;
}
bool small_fine_cut(const verify_params& P, unsigned others) {
    const bool vargen = P.eq[0].comb == nullptr;
    return P.n <= (vargen ? SMALL_PATH_FINE_ITEMS_VARGEN : SMALL_PATH_FINE_ITEMS[P.n_eq]) && others < SMALL_ECONOMY_FROM;
}
// Sixteen pieces for the smallest calls that are alone: phase B (the tail behind the hash) is two windows per lane instead of
// four, for twice the chain lanes.  One call, 16 against 8 pieces (profiles/r04_small_call_chain.jsonl): single 0.365-0.38
// against 0.40-0.41 ms up to 512 items (1 024: 0.45 against 0.42); double 0.50-0.54 against 0.525-0.56 up to 1 024; per-item
// generator 0.41-0.42 against 0.46 up to 256 (512: equal; 1 024: 0.61 against 0.53).
constexpr size_t SMALL_PATH_FINEST_ITEMS[3] = {0, 512, 1024}, SMALL_PATH_FINEST_ITEMS_VARGEN = 256;      // by number of equations
bool small_finest_size(const verify_params& P) {
#if defined(JJS_AB_NO_FINEST_CUT)
    return false;
#endif
    return P.n <= (P.eq[0].comb == nullptr ? SMALL_PATH_FINEST_ITEMS_VARGEN : SMALL_PATH_FINEST_ITEMS[P.n_eq]);
}
uint32_t small_positions(const verify_params& P, unsigned others) {
    uint32_t positions = small_fine_cut(P, others) ? (small_finest_size(P) ? 16 : 8) : 4;
#if defined(JJS_PROFILING)
    if (g_force_positions) positions = (uint32_t)g_force_positions;
#endif
    return positions;
}
size_t small_table_bytes(const verify_params& P, uint32_t positions) { return P.n * sb_table_words_per_item(P.n_eq, positions) * sizeof(uint32_t); }
// the tables of the finest cut a call of this size can take (a slot sized for them holds any coarser cut)
size_t small_table_bytes_max(const verify_params& P) { return small_table_bytes(P, small_finest_size(P) ? 16 : 8); }
int launch_small(verify_params P, hipStream_t s) {
    const bool vargen = P.eq[0].comb == nullptr;
    const unsigned others = small_calls_in_flight();
    const uint32_t positions = small_positions(P, others);
    const size_t table_bytes = small_table_bytes(P, positions);
    if (int rc = ensure_small((positions > 8 ? small_table_bytes(P, positions) : small_table_bytes_max(P)) + 4 * P.n + 64)) return rc;
    small_params S{};
    P.small_mode = 1;
    S.V = P;
    S.tables = reinterpret_cast<uint32_t*>(sl->small);
    S.point_ok = sl->small + table_bytes;
    S.positions = positions;
    S.windows = vargen ? 64 : 32;

    // The chains on four lanes each (sb_chain_lane_quad) while the call leaves lanes idle: the far positions of a per-item-
    // generator call are 224 dependent doublings (full-size scalars), which outlast the hash beside them (one call of 1 ...
    // 2 048 such signatures: 0.83-0.92 -> 0.62-0.67 ms; at 4 096 the fourfold chain lanes would be four waves per SIMD); those of
    // the fixed-generator schemes 112, which the hash outlasted until its chain was cut to three products a round
    // (hades29.h): since then 0.44-0.46 -> 0.40-0.42 ms for one call of 1 ... 1 024 single signatures
    // (profiles/r04_quad_small_chains.jsonl, r04_small_call_chain.jsonl).
    // (the double scheme gains nothing at any size, and single calls nothing beyond 1 024 items: profiles/r04_small_call_chain.jsonl)
    S.quad_chains = (small_fine_cut(P, others) && (vargen ? P.n <= SMALL_QUAD_CHAIN_MAX_ITEMS : (P.n_eq == 1 && P.n <= SMALL_QUAD_CHAIN_MAX_ITEMS_FIXED))) ? 1u : 0u;
    // eight lanes per hash where the hash is the critical path -- a fixed generator, or a per-item generator whose chains run on
    // quads -- and the batch leaves lanes idle
    // (a fixed generator: up to SMALL_PATH_COOP_ITEMS, beyond the fine cut -- 6 144 single signatures 0.58 against 0.70 ms, double
    // 0.78 against 0.93; at 8 192 the eightfold hash lanes no longer fit beside the chains: 0.76 against 0.71,
    // profiles/r04_small_call_chain.jsonl)
    S.hash_lanes = ((small_fine_cut(P, others) && (!vargen || S.quad_chains)) || (!vargen && others < SMALL_ECONOMY_FROM && P.n <= SMALL_PATH_COOP_ITEMS)) ? SB_HASH_LANES : 1;
    const unsigned hash_blocks = (unsigned)((P.n * S.hash_lanes + BLOCK - 1) / BLOCK);
    const unsigned chain_blocks = (unsigned)((P.n * P.n_eq * 2 * (S.quad_chains ? 4 : 1) + BLOCK - 1) / BLOCK);
    const unsigned point_blocks = (unsigned)((P.n * P.n_points + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(small_a_kernel, dim3(hash_blocks + positions * chain_blocks + point_blocks), dim3(BLOCK), 0, s, S,
                       (uint32_t)hash_blocks, (uint32_t)chain_blocks);
    hipLaunchKernelGGL(small_b_kernel, dim3((unsigned)((P.n * P.n_eq * positions + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, S);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
