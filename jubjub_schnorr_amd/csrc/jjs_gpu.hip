// libjjs_gpu.so: HIP kernels (gfx950) + the C ABI of include/jjs_gpu.h.
//
// A verification call takes one of three paths, chosen from its size and from how often its keys repeat (nothing
// else: no switch, no environment variable), all with the same statuses:
//   throughput  (verify_core.h)   one signature per lane: prepare_kernel (hash, scalar lattice, pairing tests; four
//               waves per SIMD), verify_kernel (the equations; a persistent pass whose grid is what the chip holds
//               resident, each lane owning WS_WORDS_PER_LANE words of workspace for its window tables) and
//               resolve_kernel (the items verify_kernel could not decide);
//   key tables  (key_tables.h)    >= 65 536 items whose keys repeat >= 16 times on average: keys deduplicated on the
//               device, validity and window tables once per key (second stream, beside the hashes), additions only
//               per signature (key_verify_kernel); decided on the device, no host round trip;
//   latency     (small_batch.h)   <= 16 384 items: one signature spread over 11-45 lanes in two launches.
// The per-status tally is reduced with wave ballots and one atomic per wave per status.  One process can drive
// several devices (jjs_init); all state is per device, per-call state lives in call slots (call_slot).
//
// This translation unit in parts (textual includes inside one anonymous namespace, in this order):
//   device_kernels.h   every __global__ entry
//   engine_state.h     device state, call slots, staging threads, errors, slot ordering, grow-only buffers
//   verify_job.h       the arenas of the key-table path; a verification call in stages (begin, ingest, keys, hash, finish)
//   (here)             device set-up and tear-down, the RCCL clique
//   host_calls.h       the large blocking host-buffer calls: upload plan, staging copies, the pipeline
//   host_lanes.h       the small ones (included among the entry points, behind the table of call shapes): staging lanes
//                      outside the engine's mutex, calls of several threads in one launch
//   (here)             the extern "C" entry points: one staged_call builder per scheme and input format
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <random>
#include <sched.h>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <thread>
#include <vector>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif
#include <rccl/rccl.h>

#include "../../include/jjs_gpu.h"
#include "schemes.h"
#include "decode.h"
#include "normalize.h"
#include "small_batch.h"
#include "key_tables.h"
#include "sign_core.h"
#include "multisig_core.h"
#include "jjs_sponge_tags_long.inc"

using namespace jjs;

namespace {

#include "device_kernels.h"
#include "engine_state.h"
#include "verify_job.h"

// Every entry point works on the calling thread's current HIP device, which must be one jjs_init set up.
int check_ready() {
    if (L.devs.empty()) return fail(JJS_ERR_NOT_INIT, "jjs_init has not been called");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    for (device_state* d : L.devs)
        if (d->device == dev) { g = d; return JJS_OK; }
    return fail(JJS_ERR_NOT_INIT, "the current HIP device (%d) is not one of the %zu this process initialised", dev,
                L.devs.size());
}

template <typename... Ptrs>
bool all_ok(Ptrs... p) {
    return ((p != nullptr && aligned16(p)) && ...);
}

int init_device(device_state& d, int ordinal) {
    d.device = ordinal;
    HIP_TRY(hipSetDevice(ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
    HIP_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    for (hipStream_t& side : d.side) HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d.host_begin, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking));
    {   // the per-key kernels are few, long waves that must finish before the challenge hashes do: dispatch them first
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        d.key_priority = hi;
        d.table_priority = lo;
    }
    HIP_TRY(hipEventCreateWithFlags(&d.side_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.ingest_done, hipEventDisableTiming));
    for (size_t i = 0; i < HOST_MAX_PIECES; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&d.chunk_up[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&d.chunk_done[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&d.last_use, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(d.last_use, d.stream));
    int per_cu_v = 0, per_cu_s = 0, per_cu_m = 0, per_cu_r = 0;
    int per_cu_p = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_p, prepare_kernel, BLOCK, 0));
    d.grid_prepare = prop.multiProcessorCount * (per_cu_p < 1 ? 1 : per_cu_p) * 4;   // not persistent: a few blocks per slot
    int per_cu_k = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_k, key_verify_kernel, BLOCK, 0));
    d.grid_key_verify = prop.multiProcessorCount * (per_cu_k < 1 ? 1 : per_cu_k) * 4;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_r, resolve_kernel, BLOCK, 0));
    d.grid_resolve = prop.multiProcessorCount * (per_cu_r < 1 ? 1 : per_cu_r);
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_v, verify_kernel, BLOCK, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_s, sign_kernel, BLOCK, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_m, msig_kernel, BLOCK, 0));
    if (per_cu_v < 1) per_cu_v = 1;
    if (per_cu_s < 1) per_cu_s = 1;
    if (per_cu_m < 1) per_cu_m = 1;
    d.grid_sign = prop.multiProcessorCount * per_cu_s;
    d.grid_msig = prop.multiProcessorCount * per_cu_m;
    // slot 0: the workspace of the largest persistent grid (verify, sign, multisig share it); small slots: one
    // lane's worth per item of the largest call they take
    d.slots[0].grid_verify = prop.multiProcessorCount * per_cu_v;
    int lanes_blocks = d.slots[0].grid_verify > d.grid_sign ? d.slots[0].grid_verify : d.grid_sign;
    if (d.grid_msig > lanes_blocks) lanes_blocks = d.grid_msig;
    for (int i = 0; i < N_SLOTS; ++i) {
        call_slot& c = d.slots[i];
        const bool big = i == 0 || i == SECOND_BIG_SLOT;
        const size_t slot_items = i <= N_SMALL_SLOTS ? SMALL_SLOT_ITEMS : MEDIUM_SLOT_ITEMS;
        c.grid_verify = big ? d.slots[0].grid_verify : (int)(slot_items / BLOCK);
        // slot 0 also serves the signer and the multisig kernels, whose grids may be larger than the verify grid
        const size_t lanes = i == 0 ? (size_t)lanes_blocks * BLOCK : (big ? (size_t)c.grid_verify * BLOCK : slot_items);
        HIP_TRY(hipMalloc(&c.workspace, lanes * WS_WORDS_PER_LANE * sizeof(uint32_t)));
        HIP_TRY(hipEventCreateWithFlags(&c.last_use, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c.last_use, d.stream));
        if (i == 0 || i > N_SMALL_SLOTS) {       // slots whose calls can be large enough for the key tables: a key stream each
            HIP_TRY(hipEventCreateWithFlags(&c.key_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.key_mid, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.key_join, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.key_ahead, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.key_chains, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.key_cleared, hipEventDisableTiming));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c.seen), sizeof(key_feedback), hipHostMallocDefault));
            memset(c.seen, 0, sizeof(key_feedback));
        }
    }
    // The priority streams, in the order of their importance: the runtime hands its high-priority hardware queues out in
    // creation order, and the key streams of the big slots must have one each (scripts/timeline.sh: created behind two
    // other priority streams, the small launches of the first big slot's key stream waited 0.3-0.5 ms each for wave slots
    // instead of 0.1, and a resident 2^20 batch took 0.5 ms longer).
    // The two streams on which a host-buffer call normalises extended points come right behind the big slots' key streams: a
    // large host call runs in the second big slot, and with its ingest stream on the hardware queue of that slot's key stream the
    // normalisation of its last range waited 1.6 ms behind the key chains and tables (scripts/host_timeline.sh ... ext; a 2^20-item
    // single call 12.1-12.3 -> 11.6-11.8 ms, profiles/r04_host_ext_ab.jsonl).
    {
        static_assert(N_MEDIUM_SLOTS == 3 && N_BIG_SLOTS == 2, "one entry per slot that has a key stream");
        for (int i : {0, SECOND_BIG_SLOT}) HIP_TRY(hipStreamCreateWithPriority(&d.slots[i].key_stream, hipStreamNonBlocking, d.key_priority));
        for (hipStream_t& is : d.ingest) HIP_TRY(hipStreamCreateWithPriority(&is, hipStreamNonBlocking, d.key_priority));
        for (int i : {1 + N_SMALL_SLOTS, 2 + N_SMALL_SLOTS, 3 + N_SMALL_SLOTS})
            HIP_TRY(hipStreamCreateWithPriority(&d.slots[i].key_stream, hipStreamNonBlocking, d.key_priority));
    }
    for (call_slot& c : d.slots)
        if (c.key_stream) HIP_TRY(hipStreamCreateWithPriority(&c.table_stream, hipStreamNonBlocking, d.table_priority));
    HIP_TRY(hipMalloc(&d.comb_g, COMB_TABLE_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d.comb_gn, COMB_TABLE_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d.tag, 32));
    HIP_TRY(hipMalloc(&d.tally, 4 * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&d.dlog_pow, DLOG_POW_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d.dlog_hash, 65536));
    HIP_TRY(hipMemsetAsync(d.dlog_hash, 0, 65536, d.stream));
    hipLaunchKernelGGL(dlog_table_kernel, dim3(7), dim3(BLOCK), 0, d.stream, d.dlog_pow, d.dlog_hash);
    HIP_TRY(hipMalloc(&d.tags_long, sizeof(JJS_SPONGE_TAG_LONG)));
    HIP_TRY(hipMemcpyAsync(d.tags_long, JJS_SPONGE_TAG_LONG, sizeof(JJS_SPONGE_TAG_LONG), hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.tag, JJS_DOUBLE_TAG_WORDS, 32, hipMemcpyHostToDevice, d.stream));
    const int blocks = (COMB_WINDOWS * COMB_ENTRIES + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(comb_kernel, dim3(blocks), dim3(BLOCK), 0, d.stream, d.comb_g, 0);
    hipLaunchKernelGGL(comb_kernel, dim3(blocks), dim3(BLOCK), 0, d.stream, d.comb_gn, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d.stream));
    return JJS_OK;
}

void free_device(device_state& d) {
    if (d.device < 0) return;
    (void)hipSetDevice(d.device);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    void* bufs[] = {d.comb_g, d.comb_gn, d.tag, d.tally, d.msig, d.tags_long, d.dlog_pow, d.dlog_hash, d.stage};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (call_slot& c : d.slots) {
        if (c.key_stream) { (void)hipStreamSynchronize(c.key_stream); (void)hipStreamDestroy(c.key_stream); }
        if (c.table_stream) { (void)hipStreamSynchronize(c.table_stream); (void)hipStreamDestroy(c.table_stream); }
        void* sb[] = {c.workspace, c.pending, c.prep, c.wire, c.small, c.keys, c.key_pool};
        for (void* b : sb)
            if (b) (void)hipFree(b);
        if (c.seen) (void)hipHostFree(c.seen);
        hipEvent_t evs[] = {c.last_use, c.key_fork, c.key_mid, c.key_join, c.key_ahead, c.key_chains, c.key_cleared};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    if (d.pinned) (void)hipHostFree(d.pinned);
    for (host_lane& l : d.lanes) {
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.dev) (void)hipFree(l.dev);
        if (l.pinned) (void)hipHostFree(l.pinned);
    }
    (void)hipDeviceSynchronize();
    free_retired(d);
    if (d.last_use) (void)hipEventDestroy(d.last_use);
    for (size_t i = 0; i < HOST_MAX_PIECES; ++i) {
        if (d.chunk_up[i]) (void)hipEventDestroy(d.chunk_up[i]);
        if (d.chunk_done[i]) (void)hipEventDestroy(d.chunk_done[i]);
    }
    if (d.copy_stream) { (void)hipStreamSynchronize(d.copy_stream); (void)hipStreamDestroy(d.copy_stream); }
    for (hipStream_t side : d.side)
        if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    for (hipStream_t is : d.ingest)
        if (is) { (void)hipStreamSynchronize(is); (void)hipStreamDestroy(is); }
    if (d.host_begin) (void)hipEventDestroy(d.host_begin);
    if (d.side_join) (void)hipEventDestroy(d.side_join);
    if (d.ingest_done) (void)hipEventDestroy(d.ingest_done);
    delete d.stagers;
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d.device = -1;
}

int load_rccl() {
    rccl_api& r = L.rccl;
    if (r.handle) return JJS_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(JJS_ERR_COLLECTIVE, "cannot load RCCL: %s", dlerror());
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(h);
        r = rccl_api{};
        return fail(JJS_ERR_COLLECTIVE, "RCCL library lacks a required symbol");
    }
    r.handle = h;
    return JJS_OK;
}

// One communicator per driven device, all in this process (ranks = positions in L.devs).
int start_comms() {
    if (int rc = load_rccl()) return rc;
    int ords[MAX_DEVICES];
    for (size_t i = 0; i < L.devs.size(); ++i) ords[i] = L.devs[i]->device;
    RCCL_TRY(L.rccl.CommInitAll(L.comms, (int)L.devs.size(), ords));
    L.comms_up = true;
    return JJS_OK;
}

// Sum of the 4-counter tallies over the driven devices, in place in every device's buffer; queued on each
// device's stream behind the launch that produced the counters (the only collective).
int allreduce_tallies() {
    RCCL_TRY(L.rccl.GroupStart());
    for (size_t i = 0; i < L.devs.size(); ++i) {
        device_state& d = *L.devs[i];
        ncclResult_t r = L.rccl.AllReduce(d.tally, d.tally, 4, ncclUint64, ncclSum, L.comms[i], d.stream);
        if (r != ncclSuccess) {
            (void)L.rccl.GroupEnd();
            return fail(JJS_ERR_COLLECTIVE, "ncclAllReduce: %s", L.rccl.GetErrorString(r));
        }
    }
    RCCL_TRY(L.rccl.GroupEnd());
    return JJS_OK;
}

void shutdown_locked() {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (device_state* d : L.devs)
        if (d->stream) { (void)hipSetDevice(d->device); (void)hipStreamSynchronize(d->stream); }
    if (L.comms_up) {
        for (size_t i = 0; i < L.devs.size(); ++i) (void)L.rccl.CommDestroy(L.comms[i]);
        L.comms_up = false;
    }
    for (device_state* d : L.devs) { free_device(*d); delete d; }
    L.devs.clear();
    L.virtual_devices = false;
    g = nullptr;
    if (prev >= 0) (void)hipSetDevice(prev);
}

struct device_restore {   // puts the calling thread back on the device it came in with
    int prev = -1;
    device_restore() { (void)hipGetDevice(&prev); }
    ~device_restore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#include "host_calls.h"

}  // namespace

extern "C" {

#if defined(JJS_LANE_TRACE)
static void lane_trace_dump();     // host_lanes.h, investigation builds
#endif
int jjs_abi_version(void) { return 5; }
const char* jjs_last_error(void) { return t_err; }

int jjs_init(int device_count) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (device_count < 0 || device_count > MAX_DEVICES)
        return fail(JJS_ERR_ARG, "device_count must be in [0, %d] (got %d)", MAX_DEVICES, device_count);
    int visible = 0, current = -1;
    HIP_TRY(hipGetDeviceCount(&visible));
    HIP_TRY(hipGetDevice(&current));
    if (visible < 1) return fail(JJS_ERR_HIP, "no HIP device visible");
#if defined(JJS_PROFILING)
    const bool allow_virtual = g_allow_virtual;   // logical devices sharing one card: test builds only
#else
    const bool allow_virtual = false;
#endif
    const int want = device_count == 0 ? visible : device_count;
    if (!L.devs.empty()) {   // idempotent for the same request
        if (device_count == 1) return check_ready();
        if ((size_t)want == L.devs.size()) return JJS_OK;
        return fail(JJS_ERR_ARG, "already initialised with %zu device(s); call jjs_shutdown first", L.devs.size());
    }
    if (want > visible && !allow_virtual)
        return fail(JJS_ERR_ARG, "device_count=%d but only %d HIP device(s) are visible", want, visible);
    L.virtual_devices = want > visible;
    for (int i = 0; i < want; ++i) {
        device_state* d = new device_state();
        L.devs.push_back(d);
        int rc = init_device(*d, device_count == 1 ? current : i % visible);
        if (rc) { shutdown_locked(); (void)hipSetDevice(current); return rc; }
    }
    HIP_TRY(hipSetDevice(current));
    if (L.devs.size() > 1 && !L.virtual_devices) {
        int rc = start_comms();
        if (rc) { shutdown_locked(); return rc; }
    }
    return JJS_OK;
}

void jjs_shutdown(void) {
    {   // a large host-buffer call in progress finishes first (it holds its device's host_mu, not the engine's mutex)
        std::vector<device_state*> devs;
        { std::lock_guard<std::mutex> lock(L.mu); devs = L.devs; }
        for (device_state* d : devs) { std::lock_guard<std::mutex> big(d->host_mu); }
    }
    std::unique_lock<std::mutex> lock(L.mu);
    // host-buffer calls that hold a lane finish first (they wait for the device outside the mutex)
    L.lane_cv.wait(lock, [] {
        for (device_state* d : L.devs)
            for (const host_lane& l : d->lanes)
                if (l.state != host_lane::FREE) return false;
        return true;
    });
    shutdown_locked();
    L.lane_cv.notify_all();
#if defined(JJS_LANE_TRACE)
    lane_trace_dump();
#endif
}

int jjs_device_count(void) {
    std::lock_guard<std::mutex> lock(L.mu);
    return (int)L.devs.size();
}

int jjs_collective_ranks(void) {
    std::lock_guard<std::mutex> lock(L.mu);
    return L.comms_up ? (int)L.devs.size() : 0;
}

int jjs_stream_sync(void* stream) {
    {
        std::lock_guard<std::mutex> lock(L.mu);
        if (L.devs.empty()) return fail(JJS_ERR_NOT_INIT, "jjs_init has not been called");
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));     // waits without holding the engine's mutex
    return JJS_OK;
}

// ---- the staged_call of every scheme and input format --------------------------------------------------------------
// A builder picks the call slot, sizes what the format needs in it and fills in the descriptors from the device arrays
// d[0..] (in the order of the entry point's arguments).  Resident calls launch it at once (launch_staged), host-buffer
// calls feed it piece by piece (run_host_block).
static int ensure_wire(size_t n) {
    if (n <= sl->wire_items) return JJS_OK;
    const size_t cap = grown(n < 4096 ? 4096 : n);
    return regrow(sl->wire, sl->wire_items, sl->wire_items * (4 * 64 + 16 + 2 * 48), cap, cap * (4 * 64 + 16 + 2 * 48));
}
static uint8_t* wire_pts(int k) { return sl->wire + (size_t)k * sl->wire_items * 64; }
static uint8_t* wire_bad() { return sl->wire + (size_t)4 * sl->wire_items * 64; }
// prefix products of the normalisation (normalize.h): one area for the key columns and one for the others, whose
// launches may overlap in a host-buffer call
static uint32_t* wire_scratch(int k) { return reinterpret_cast<uint32_t*>(sl->wire + (size_t)sl->wire_items * (4 * 64 + 16 + 48 * k)); }

static int build_affine_single(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_single((const uint8_t*)d[0], (const uint8_t*)d[1], (const uint8_t*)d[2], (const uint8_t*)d[3], n, g->comb_g, o);
    return JJS_OK;
}
static int build_affine_double(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3], d[4], d[5])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_double((const uint8_t*)d[0], (const uint8_t*)d[1], (const uint8_t*)d[2], (const uint8_t*)d[3], (const uint8_t*)d[4],
                        (const uint8_t*)d[5], n, g->tag, g->comb_g, g->comb_gn, o);
    return JJS_OK;
}
static int build_affine_vargen(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3], d[4])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_vargen((const uint8_t*)d[0], (const uint8_t*)d[1], (const uint8_t*)d[2], (const uint8_t*)d[3], (const uint8_t*)d[4], n, o);
    return JJS_OK;
}

// wire formats: d = sig, pk, m.  The R points are decoded per item into wire_pts(0) (1), the keys per key or per item
// (job_keys / job_hash) into the columns behind them; the flags of rejected encodings are P.pre_malformed.
static int wire_common(staged_call& C, const void* sig, uint32_t sig_stride, uint32_t n_r, size_t n) {
    decode_params D{};
    D.n_src = n_r; D.n = n; D.bad = wire_bad();
    for (uint32_t k = 0; k < n_r; ++k) { D.src[k] = fe_src{(const uint8_t*)sig, sig_stride, 32 + 32 * k}; D.out[k] = wire_pts((int)k); }
    C.wire = true;
    C.W.sig = D;
    C.W.bad = wire_bad();
    C.P.u = fe_src{(const uint8_t*)sig, sig_stride, 0};
    C.P.pre_malformed = wire_bad();
    C.P.decoded_points = 1;
    return JJS_OK;
}
static int build_wire_single(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_single((const uint8_t*)d[0], wire_pts(0), wire_pts(1), (const uint8_t*)d[2], n, g->comb_g, o);
    C.W.n_cols = 1;
    C.W.comp[0] = fe_src{(const uint8_t*)d[1], 32, 0};  C.W.out[0] = wire_pts(1);      // PK
    return wire_common(C, d[0], 64, 1, n);
}
static int build_wire_double(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_double((const uint8_t*)d[0], wire_pts(0), wire_pts(1), wire_pts(2), wire_pts(3), (const uint8_t*)d[2], n, g->tag,
                        g->comb_g, g->comb_gn, o);
    C.W.n_cols = 2;
    C.W.comp[0] = fe_src{(const uint8_t*)d[1], 64, 0};  C.W.out[0] = wire_pts(2);      // PK
    C.W.comp[1] = fe_src{(const uint8_t*)d[1], 64, 32}; C.W.out[1] = wire_pts(3);      // PK'
    return wire_common(C, d[0], 96, 2, n);
}
static int build_wire_vargen(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_vargen((const uint8_t*)d[0], wire_pts(0), wire_pts(1), wire_pts(2), (const uint8_t*)d[2], n, o);
    C.W.n_cols = 2;
    C.W.comp[0] = fe_src{(const uint8_t*)d[1], 64, 0};  C.W.out[0] = wire_pts(1);      // PK
    C.W.comp[1] = fe_src{(const uint8_t*)d[1], 64, 32}; C.W.out[1] = wire_pts(2);      // generator
    return wire_common(C, d[0], 64, 1, n);
}

// extended coordinates (U, V, Z): normalised on the device into wire_pts(k), then the affine descriptors.  `keys`: bit k
// set when point column k is a key column (they arrive, and are normalised, ahead of the others in a host-buffer call).
static void ext_common(staged_call& C, const void* const* pts, uint32_t n_pts, uint32_t keys) {
    C.ext = true;
    for (uint32_t grp = COLS_KEYS; grp <= COLS_ALL; ++grp) {
        normalize_params& N = C.N[grp];
        N = normalize_params{};
        for (uint32_t k = 0; k < n_pts; ++k) {
            const bool is_key = ((keys >> k) & 1u) != 0;
            if (!((is_key ? COLS_KEYS : COLS_REST) & grp)) continue;
            N.src[N.n_src] = fe_src{(const uint8_t*)pts[k], 96, 0};
            N.out[N.n_src] = wire_pts((int)k);
            ++N.n_src;
        }
        N.bad = wire_bad();
        N.scratch = wire_scratch(grp == COLS_KEYS ? 1 : 0);
    }
    C.P.pre_malformed = wire_bad();
}
static int build_ext_single(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_single((const uint8_t*)d[0], wire_pts(0), wire_pts(1), (const uint8_t*)d[3], n, g->comb_g, o);
    const void* pts[] = {d[1], d[2]};                      // R, PK
    ext_common(C, pts, 2, 2u);
    return JJS_OK;
}
static int build_ext_double(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3], d[4], d[5])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_double((const uint8_t*)d[0], wire_pts(0), wire_pts(1), wire_pts(2), wire_pts(3), (const uint8_t*)d[5], n, g->tag,
                        g->comb_g, g->comb_gn, o);
    const void* pts[] = {d[1], d[2], d[3], d[4]};          // R, R', PK, PK'
    ext_common(C, pts, 4, 12u);
    return JJS_OK;
}
static int build_ext_vargen(const void* const* d, size_t n, void* status, void* tally, hipStream_t s, staged_call& C) {
    if (n && !all_ok(d[0], d[1], d[2], d[3], d[4])) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    pick_slot(n, s);
    if (int rc = ensure_wire(n)) return rc;
    out_ptrs o{(uint8_t*)status, (unsigned long long*)tally, nullptr, sl->workspace};
    C = staged_call{};
    C.P = params_vargen((const uint8_t*)d[0], wire_pts(0), wire_pts(1), wire_pts(2), (const uint8_t*)d[4], n, o);
    const void* pts[] = {d[1], d[2], d[3]};                // R, PK, Gen
    ext_common(C, pts, 3, 6u);
    return JJS_OK;
}

// a resident call: device pointers d[], asynchronous on `stream`
static int resident_call(call_builder build, const void* const* d, size_t n, void* status, void* tally, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (status && !aligned16(status)) return fail(JJS_ERR_ARG, "status must be 16-byte aligned");
    if (n == 0) {
        if (tally) HIP_TRY(hipMemsetAsync(tally, 0, 4 * sizeof(unsigned long long), s));
        return JJS_OK;
    }
    staged_call C;
    if (int rc = build(d, n, status, tally, s, C)) return rc;
    return launch_staged(C, s);
}

// The shape of every verification entry point: its builder and the columns of a host-buffer call in the order of the
// entry point's arguments (width in bytes, and the group that decides when a large call uploads the column: host_calls.h).
struct call_shape {
    call_builder build;
    size_t n_cols;
    struct { size_t width; uint32_t group; } col[8];
    int wire_points;              // R points per signature that a wire call decodes per item
};
static const call_shape SHAPES[3][3] = {       // [JJS_SCHEME_*][JJS_FORMAT_*]
    {{build_affine_single, 4, {{32, COLS_LATE}, {64, COLS_REST}, {64, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_ext_single, 4, {{32, COLS_LATE}, {96, COLS_REST}, {96, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_wire_single, 3, {{64, COLS_REST}, {32, COLS_KEYS}, {32, COLS_REST}}, 1}},
    {{build_affine_double, 6, {{32, COLS_LATE}, {64, COLS_REST}, {64, COLS_REST}, {64, COLS_KEYS}, {64, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_ext_double, 6, {{32, COLS_LATE}, {96, COLS_REST}, {96, COLS_REST}, {96, COLS_KEYS}, {96, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_wire_double, 3, {{96, COLS_REST}, {64, COLS_KEYS}, {32, COLS_REST}}, 2}},
    {{build_affine_vargen, 5, {{32, COLS_LATE}, {64, COLS_REST}, {64, COLS_KEYS}, {64, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_ext_vargen, 5, {{32, COLS_LATE}, {96, COLS_REST}, {96, COLS_KEYS}, {96, COLS_KEYS}, {32, COLS_REST}}, 0},
     {build_wire_vargen, 3, {{64, COLS_REST}, {64, COLS_KEYS}, {32, COLS_REST}}, 1}},
};

#include "host_lanes.h"

// a host-buffer call: blocking
static int host_call(int scheme, int format, const uint8_t* const* ptrs, size_t n, uint8_t* status, uint64_t tally[4]) {
    const call_shape& S = SHAPES[scheme][format];
    for (size_t k = 0; k < S.n_cols; ++k)
        if (n && !ptrs[k]) return fail(JJS_ERR_ARG, "null input pointer");
    bool one_device = false;
    {
        std::lock_guard<std::mutex> lock(L.mu);
        if (int rc = check_ready()) return rc;
        one_device = L.devs.size() == 1;
    }
    if (n == 0) {
        if (tally) for (int k = 0; k < 4; ++k) tally[k] = 0;
        return JJS_OK;
    }
    if (one_device && n <= LANE_MAX_ITEMS) return lane_call(scheme, format, ptrs, n, status, tally);
    host_col cols[8];
    for (size_t k = 0; k < S.n_cols; ++k) cols[k] = host_col{ptrs[k], S.col[k].width, S.col[k].group};
    if (one_device) {
        // A large call fills the device by itself: such calls run one at a time per device (host_mu; they share the device's
        // staging), but outside the engine's mutex, which they take only to pick their slot -- other threads' calls are queued
        // meanwhile.  (jjs_shutdown and jjs_trim take host_mu, too.)
        device_state* dev = nullptr;
        {
            std::lock_guard<std::mutex> lock(L.mu);
            if (int rc = check_ready()) return rc;
            dev = g;
        }
        std::lock_guard<std::mutex> big(dev->host_mu);
        {
            std::lock_guard<std::mutex> lock(L.mu);
            if (L.devs.empty() || check_ready() != JJS_OK || g != dev) return fail(JJS_ERR_NOT_INIT, "the engine was shut down during the call");
        }
        return no_throw([&] { return run_host(cols, S.n_cols, n, status, tally, S.build, S.wire_points, true); });
    }
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    return no_throw([&] { return run_host(cols, S.n_cols, n, status, tally, S.build, S.wire_points); });
}

// ---- affine inputs: device-buffer and host-buffer entry points ----------------------------------------------
int jjs_verify_single_dev(const void* u, const void* R, const void* PK, const void* m, size_t n, void* status,
                          void* tally, void* stream) {
    const void* d[] = {u, R, PK, m};
    return resident_call(build_affine_single, d, n, status, tally, stream);
}
int jjs_verify_double_dev(const void* u, const void* R, const void* Rp, const void* PK, const void* PKp, const void* m,
                          size_t n, void* status, void* tally, void* stream) {
    const void* d[] = {u, R, Rp, PK, PKp, m};
    return resident_call(build_affine_double, d, n, status, tally, stream);
}
int jjs_verify_vargen_dev(const void* u, const void* R, const void* PK, const void* Gen, const void* m, size_t n,
                          void* status, void* tally, void* stream) {
    const void* d[] = {u, R, PK, Gen, m};
    return resident_call(build_affine_vargen, d, n, status, tally, stream);
}
int jjs_verify_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n,
                      uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, PK, m};
    return host_call(JJS_SCHEME_SINGLE, JJS_FORMAT_AFFINE, p, n, status, tally);
}
int jjs_verify_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                      const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, Rp, PK, PKp, m};
    return host_call(JJS_SCHEME_DOUBLE, JJS_FORMAT_AFFINE, p, n, status, tally);
}
int jjs_verify_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                      size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, PK, Gen, m};
    return host_call(JJS_SCHEME_VARGEN, JJS_FORMAT_AFFINE, p, n, status, tally);
}

// ---- wire formats: on-device decoding, then the same verify kernels -----------------------------------
int jjs_verify_single_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally, void* stream) {
    const void* d[] = {sig, pk, m};
    return resident_call(build_wire_single, d, n, status, tally, stream);
}
int jjs_verify_double_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally, void* stream) {
    const void* d[] = {sig, pk, m};
    return resident_call(build_wire_double, d, n, status, tally, stream);
}
int jjs_verify_vargen_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally, void* stream) {
    const void* d[] = {sig, pk, m};
    return resident_call(build_wire_vargen, d, n, status, tally, stream);
}
int jjs_verify_single_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {sig, pk, m};
    return host_call(JJS_SCHEME_SINGLE, JJS_FORMAT_WIRE, p, n, status, tally);
}
int jjs_verify_double_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {sig, pk, m};
    return host_call(JJS_SCHEME_DOUBLE, JJS_FORMAT_WIRE, p, n, status, tally);
}
int jjs_verify_vargen_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {sig, pk, m};
    return host_call(JJS_SCHEME_VARGEN, JJS_FORMAT_WIRE, p, n, status, tally);
}

// ---- extended coordinates (U, V, Z): normalised on the device, then the same verify kernels -----------------
int jjs_verify_single_ext_dev(const void* u, const void* R, const void* PK, const void* m, size_t n, void* status, void* tally,
                              void* stream) {
    const void* d[] = {u, R, PK, m};
    return resident_call(build_ext_single, d, n, status, tally, stream);
}
int jjs_verify_double_ext_dev(const void* u, const void* R, const void* Rp, const void* PK, const void* PKp, const void* m,
                              size_t n, void* status, void* tally, void* stream) {
    const void* d[] = {u, R, Rp, PK, PKp, m};
    return resident_call(build_ext_double, d, n, status, tally, stream);
}
int jjs_verify_vargen_ext_dev(const void* u, const void* R, const void* PK, const void* Gen, const void* m, size_t n,
                              void* status, void* tally, void* stream) {
    const void* d[] = {u, R, PK, Gen, m};
    return resident_call(build_ext_vargen, d, n, status, tally, stream);
}
int jjs_verify_single_ext(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n, uint8_t* status,
                          uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, PK, m};
    return host_call(JJS_SCHEME_SINGLE, JJS_FORMAT_EXT, p, n, status, tally);
}
int jjs_verify_double_ext(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                          const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, Rp, PK, PKp, m};
    return host_call(JJS_SCHEME_DOUBLE, JJS_FORMAT_EXT, p, n, status, tally);
}
int jjs_verify_vargen_ext(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m, size_t n,
                          uint8_t* status, uint64_t tally[4]) {
    const uint8_t* p[] = {u, R, PK, Gen, m};
    return host_call(JJS_SCHEME_VARGEN, JJS_FORMAT_EXT, p, n, status, tally);
}

// ---- pre-sizing and trimming ------------------------------------------------------------------------------------
// What a call of this shape and size would allocate on first use, allocated now, in EVERY slot (and lane) such a call can
// land in: a service calls this once per call shape at start-up and no later call of at most that size allocates.
int jjs_reserve(int scheme, int format, size_t n_items, int host_buffers) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (scheme < 0 || scheme > 2 || format < 0 || format > 2) return fail(JJS_ERR_ARG, "scheme / format out of range");
    if (n_items == 0) return JJS_OK;
    const call_shape& S = SHAPES[scheme][format];
    struct unforce { ~unforce() { forced_slot = nullptr; } } unforce_on_every_way_out;
    return no_throw([&]() -> int {
        device_restore restore;
        std::vector<device_state*> targets;
        if (L.devs.size() == 1 || !host_buffers) targets.push_back(g); else targets = L.devs;
        size_t per = host_buffers ? (n_items + targets.size() - 1) / targets.size() : n_items;
        // small host-buffer calls combine: the launch their slot sees may carry up to COMBINE_CAP_ITEMS items
        const size_t call_items = per;
        if (host_buffers && targets.size() == 1 && per <= COMBINE_MAX_CALL_ITEMS) per = COMBINE_CAP_ITEMS;
        for (device_state* d : targets) {
            g = d;
            HIP_TRY(hipSetDevice(d->device));
            int lo, hi;
            if (per <= SMALL_SLOT_ITEMS) { lo = 1; hi = N_SMALL_SLOTS; }
            else if (per <= MEDIUM_SLOT_ITEMS) { lo = 1 + N_SMALL_SLOTS; hi = N_SMALL_SLOTS + N_MEDIUM_SLOTS; }
            else { lo = 0; hi = SECOND_BIG_SLOT; }
            for (int i = lo; i <= hi; ++i) {
                if (lo == 0 && i != 0 && i != SECOND_BIG_SLOT) continue;
                forced_slot = &d->slots[i];
                // the builder sizes what the format needs in the slot; its descriptors (built from a placeholder address,
                // never dereferenced) tell which path the call would take
                const void* in[8];
                for (size_t k = 0; k < S.n_cols; ++k) in[k] = reinterpret_cast<const void*>(uintptr_t(4096));
                staged_call C;
                int rc = S.build(in, per, nullptr, nullptr, nullptr, C);
                forced_slot = nullptr;
                if (rc) return rc;
                if (int r = reserve_for(C.P)) return r;
            }
            if (host_buffers) {
                if (targets.size() == 1 && per <= LANE_MAX_ITEMS) {
                    const lane_layout Y = lane_layout_for(S, lane_cap_for(nullptr, scheme, format, call_items));
                    for (host_lane& l : d->lanes)
                        if (int r = ensure_lane(l, Y.total)) return r;
                } else {
                    host_col cols[8];
                    for (size_t k = 0; k < S.n_cols; ++k) cols[k] = host_col{nullptr, S.col[k].width, S.col[k].group};
                    if (int r = reserve_host_block(cols, S.n_cols, per, S.wire_points)) return r;
                }
            }
        }
        g = targets[0];
        return JJS_OK;
    });
}
// Waits for the devices to go idle, then frees what growth has retired and every slot's key-table pool (a later call that
// takes the key tables allocates its pool again, at the size the slot had learnt).
int jjs_trim(void) {
    std::vector<device_state*> devs;
    {
        std::lock_guard<std::mutex> lock(L.mu);
        if (int rc = check_ready()) return rc;
        devs = L.devs;
    }
    std::vector<std::unique_lock<std::mutex>> big;            // no large host-buffer call in progress on any device
    for (device_state* d : devs) big.emplace_back(d->host_mu);
    std::unique_lock<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (devs != L.devs) return fail(JJS_ERR_NOT_INIT, "the engine was re-initialised during the call");
    device_restore restore;
    for (device_state* d : L.devs) {
        HIP_TRY(hipSetDevice(d->device));
        HIP_TRY(hipDeviceSynchronize());
        free_retired(*d);
        for (call_slot& c : d->slots) {
            if (!c.key_pool) continue;
            HIP_TRY(hipFree(c.key_pool));
            if (c.key_pool_bytes > c.key_pool_want) c.key_pool_want = c.key_pool_bytes;
            c.key_pool = nullptr; c.key_pool_bytes = 0;
        }
    }
    return JJS_OK;
}
int jjs_memory_stats(uint64_t out[JJS_MEMORY_STATS]) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (!out) return fail(JJS_ERR_ARG, "null pointer");
    uint64_t pool = 0, slots = 0, lanes = 0;
    for (const call_slot& c : g->slots) {
        pool += c.key_pool_bytes;
        slots += c.pending_items * 8 + c.prep_items * 65 + c.wire_items * (4 * 64 + 16 + 2 * 48) + c.small_bytes + c.keys_bytes;
    }
    for (const host_lane& l : g->lanes) lanes += l.dev_bytes + l.pinned_bytes;
    out[JJS_MEMORY_KEY_POOLS] = pool;
    out[JJS_MEMORY_SLOT_BUFFERS] = slots;
    out[JJS_MEMORY_HOST_STAGING] = lanes + g->stage_bytes + g->pinned_bytes;
    out[JJS_MEMORY_RETIRED] = g->retired_bytes;
    return JJS_OK;
}

// Which method the calls on the current device took (include/jjs_gpu.h).  Calls that tried the key tables are counted
// when the library next looks at their slot (their decision is made on the device): after the stream has drained,
// every finished call is in.
int jjs_path_stats(uint64_t out[JJS_PATH_STATS]) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (!out) return fail(JJS_ERR_ARG, "null pointer");
    size_t pool = 0;
    for (call_slot& c : g->slots) {
        sl = &c;
        if (c.key_stream && !c.host_owned) note_key_feedback();     // (a slot a large host-buffer call is feeding: counted by that call's successor)
        pool += c.key_pool_bytes;
    }
    for (int i = 0; i < JJS_PATH_STATS; ++i) out[i] = g->stats[i];
    out[JJS_PATH_KEY_POOL_BYTES] = pool;
    return JJS_OK;
}

static int launch_decode(decode_params D, hipStream_t s) {
    D.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
    if (int rc = begin_shared(s)) return rc;
    size_t blocks = (D.n + BLOCK - 1) / BLOCK;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, s, D);
    HIP_TRY(hipGetLastError());
    return end_shared(s);
}

int jjs_decompress_dev(const void* in, size_t n, void* affine_out, void* ok_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!all_ok(in, affine_out) || !ok_out) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    decode_params D{};
    D.n_src = 1; D.n = n; D.ok = (uint8_t*)ok_out;
    D.src[0] = fe_src{(const uint8_t*)in, 32, 0}; D.out[0] = (uint8_t*)affine_out;
    big_slot();
    return launch_decode(D, (hipStream_t)stream);
}
int jjs_compress_dev(const void* affine, size_t n, void* out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!all_ok(affine, out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipLaunchKernelGGL(compress_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t*)affine, (uint64_t)n, (uint8_t*)out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}

// ---- multisig: batch verify_share / combine -------------------------------------------
int jjs_multisig_combine_dev(const void* z, const void* PK, const void* R, const void* S, const void* m,
                             const uint32_t* offsets_host, size_t n_transcripts, void* share_status, void* transcript_status,
                             void* agg_pk, void* sig_u, void* sig_R, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n_transcripts == 0) return JJS_OK;
    if (!offsets_host || offsets_host[0] != 0) return fail(JJS_ERR_ARG, "offsets must start at 0");
    // A transcript takes any number of participants: none is the reference's InvalidMultisigTranscript (status 5 for that
    // transcript, the others are not affected), and beyond the JJS_MSIG_MAX_PARTICIPANTS the tag table covers the two
    // SAFE tags of the transcript are computed by the first pass on the device (csrc/safe_tag.h).
    for (size_t t = 0; t < n_transcripts; ++t) {
        if (offsets_host[t + 1] < offsets_host[t]) return fail(JJS_ERR_ARG, "transcript %zu: offsets must not decrease", t);
        const uint64_t cnt = offsets_host[t + 1] - offsets_host[t];
        if (cnt > JJS_MSIG_PARTICIPANTS_LIMIT)
            return fail(JJS_ERR_ARG, "transcript %zu: %llu participants (the hash transcripts are indexed with 32 bits: at most %u)", t,
                        (unsigned long long)cnt, (unsigned)JJS_MSIG_PARTICIPANTS_LIMIT);
    }
    const size_t n = offsets_host[n_transcripts];
    if ((n && !all_ok(z, PK, R, S)) || !all_ok(m, agg_pk, sig_u, sig_R) || (n && !share_status)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n > g->msig_items || n_transcripts > g->msig_transcripts) {
        size_t ci = grown(n < 4096 ? 4096 : n), ct = grown(n_transcripts < 1024 ? 1024 : n_transcripts);
        if (ci < g->msig_items) ci = g->msig_items;
        if (ct < g->msig_transcripts) ct = g->msig_transcripts;
        uint8_t* fresh = nullptr;
        HIP_TRY(hipMalloc(&fresh, ci * 4 * (1 + 8 + 2 * EXT_WORDS) + ct * 4 * (16 + 1 + 18) + 64));
        retire(g->msig, false, g->msig_items * 4 * (1 + 8 + 2 * EXT_WORDS) + g->msig_transcripts * 4 * (16 + 1 + 18) + 64);
        g->msig = fresh;
        g->msig_items = ci; g->msig_transcripts = ct;
    }
    msig_params P{};
    P.z = (const uint8_t*)z; P.PK = (const uint8_t*)PK; P.R = (const uint8_t*)R; P.S = (const uint8_t*)S; P.m = (const uint8_t*)m;
    P.n_transcripts = (uint32_t)n_transcripts; P.n_total = n;
    P.share_status = (uint8_t*)share_status; P.agg_pk = (uint8_t*)agg_pk; P.sig_u = (uint8_t*)sig_u; P.sig_R = (uint8_t*)sig_R;
    P.transcript_status = (uint8_t*)transcript_status;
    uint32_t* w = (uint32_t*)g->msig;
    P.tr_of = w; w += g->msig_items;
    P.d_words = w; w += 8 * g->msig_items;
    P.dpk = w; w += EXT_WORDS * g->msig_items;
    P.e_pt = w; w += EXT_WORDS * g->msig_items;
    P.a_words = w; w += 8 * g->msig_transcripts;
    P.c_words = w; w += 8 * g->msig_transcripts;
    uint32_t* d_off = w; w += g->msig_transcripts + 1;
    uint32_t* d_long = w;
    P.offsets = d_off;
    P.tags = g->tags_long; P.comb_g = g->comb_g; P.lane_ws = g->slots[0].workspace;
    P.max_table_participants = JJS_MSIG_MAX_PARTICIPANTS;
    P.long_tags = d_long;
    big_slot();
    if (int rc = begin_shared(s)) return rc;
    auto queue = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_off, offsets_host, (n_transcripts + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        for (int pass = 0; pass < 7; ++pass) {
            const size_t count = (pass == 0 || pass == 2 || pass == 4 || pass == 6) ? n_transcripts : n;
            // a pass with a hash chain and few items: eight lanes per item (multisig_core.h hash_lanes)
            P.hash_lanes = ((pass == 1 || pass == 2 || pass == 4) && count <= MSIG_COOP_MAX_ITEMS) ? 8u : 1u;
            hipLaunchKernelGGL(msig_kernel, dim3(grid_for(g->grid_msig, count * P.hash_lanes)), dim3(BLOCK), 0, s, P, pass);
        }
        HIP_TRY(hipGetLastError());
        return JJS_OK;
    };
    const int rc = queue();
    const int rc2 = end_shared(s);              // the slot's event covers whatever was queued, also when a step failed
    return rc ? rc : rc2;
}

// ---- challenge export ---------------------------------------------------------------------------
static int launch_challenge(challenge_params P, void* stream) {
    if (P.n == 0) return JJS_OK;
    if (!P.c_out || !aligned16(P.c_out)) return fail(JJS_ERR_ARG, "c_out null or misaligned");
    hipStream_t s = (hipStream_t)stream;
    size_t blocks = (P.n + BLOCK - 1) / BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(challenge_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, s, P);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
int jjs_challenge_single_dev(const void* R, const void* PK, const void* m, size_t n, void* c_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(R, PK, m)) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    verify_params V = params_single((const uint8_t*)m, (const uint8_t*)R, (const uint8_t*)PK, (const uint8_t*)m, n, nullptr, out_ptrs{});
    challenge_params P{};
    P.n_hash = V.n_hash; P.n = n; P.c_out = (uint8_t*)c_out;
    for (uint32_t i = 0; i < V.n_hash; ++i) P.hash_in[i] = V.hash_in[i];
    return launch_challenge(P, stream);
}
int jjs_challenge_double_dev(const void* R, const void* Rp, const void* PK, const void* PKp, const void* m, size_t n,
                             void* c_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(R, Rp, PK, PKp, m)) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    verify_params V = params_double((const uint8_t*)m, (const uint8_t*)R, (const uint8_t*)Rp, (const uint8_t*)PK,
                                    (const uint8_t*)PKp, (const uint8_t*)m, n, g->tag, nullptr, nullptr, out_ptrs{});
    challenge_params P{};
    P.n_hash = V.n_hash; P.n = n; P.c_out = (uint8_t*)c_out;
    for (uint32_t i = 0; i < V.n_hash; ++i) P.hash_in[i] = V.hash_in[i];
    return launch_challenge(P, stream);
}
int jjs_challenge_vargen_dev(const void* R, const void* PK, const void* Gen, const void* m, size_t n, void* c_out,
                             void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(R, PK, Gen, m)) return fail(JJS_ERR_ARG, "null or misaligned input pointer");
    verify_params V = params_vargen((const uint8_t*)m, (const uint8_t*)R, (const uint8_t*)PK, (const uint8_t*)Gen,
                                    (const uint8_t*)m, n, out_ptrs{});
    challenge_params P{};
    P.n_hash = V.n_hash; P.n = n; P.c_out = (uint8_t*)c_out;
    for (uint32_t i = 0; i < V.n_hash; ++i) P.hash_in[i] = V.hash_in[i];
    return launch_challenge(P, stream);
}

// ---- signing (input generator) ------------------------------------------------------------------
static int launch_sign(sign_params P, void* stream) {
    if (P.n == 0) return JJS_OK;
    hipStream_t s = (hipStream_t)stream;
    P.comb_g = g->comb_g; P.comb_gn = g->comb_gn; P.workspace = g->slots[0].workspace;
    big_slot();
    if (int rc = begin_shared(s)) return rc;
    hipLaunchKernelGGL(sign_kernel, dim3(grid_for(g->grid_sign, P.n)), dim3(BLOCK), 0, s, P);
    HIP_TRY(hipGetLastError());
    return end_shared(s);
}
int jjs_sign_single_dev(const void* sk, const void* rnd, const void* m, size_t n, void* u_out, void* R_out, void* PK_out,
                        void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(sk, rnd, m, u_out, R_out, PK_out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    sign_params P{};
    P.scheme = SCHEME_SINGLE; P.sk = (const uint8_t*)sk; P.rnd = (const uint8_t*)rnd; P.m = (const uint8_t*)m; P.n = n;
    P.u_out = (uint8_t*)u_out; P.R_out = (uint8_t*)R_out; P.PK_out = (uint8_t*)PK_out;
    return launch_sign(P, stream);
}
int jjs_sign_double_dev(const void* sk, const void* rnd, const void* m, size_t n, void* u_out, void* R_out, void* Rp_out,
                        void* PK_out, void* PKp_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(sk, rnd, m, u_out, R_out, Rp_out, PK_out, PKp_out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    sign_params P{};
    P.scheme = SCHEME_DOUBLE; P.sk = (const uint8_t*)sk; P.rnd = (const uint8_t*)rnd; P.m = (const uint8_t*)m; P.n = n;
    P.u_out = (uint8_t*)u_out; P.R_out = (uint8_t*)R_out; P.Rp_out = (uint8_t*)Rp_out; P.PK_out = (uint8_t*)PK_out;
    P.PKp_out = (uint8_t*)PKp_out;
    return launch_sign(P, stream);
}
int jjs_sign_vargen_dev(const void* sk, const void* gen_scalar, const void* rnd, const void* m, size_t n, void* u_out,
                        void* R_out, void* PK_out, void* Gen_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n && !all_ok(sk, gen_scalar, rnd, m, u_out, R_out, PK_out, Gen_out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    sign_params P{};
    P.scheme = SCHEME_VARGEN; P.sk = (const uint8_t*)sk; P.gen_scalar = (const uint8_t*)gen_scalar;
    P.rnd = (const uint8_t*)rnd; P.m = (const uint8_t*)m; P.n = n;
    P.u_out = (uint8_t*)u_out; P.R_out = (uint8_t*)R_out; P.PK_out = (uint8_t*)PK_out; P.Gen_out = (uint8_t*)Gen_out;
    return launch_sign(P, stream);
}

int jjs_public_keys_dev(const void* sk, size_t n, void* PK_out, void* PKp_out, void* bad_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!all_ok(sk, PK_out) || (PKp_out && !aligned16(PKp_out))) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    size_t blocks = (n + BLOCK - 1) / BLOCK;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(derive_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, (hipStream_t)stream, (const uint8_t*)sk,
                       (uint64_t)n, g->comb_g, g->comb_gn, (uint8_t*)PK_out, (uint8_t*)PKp_out, (uint8_t*)bad_out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}

// ---- debug primitives ------------------------------------------------------------------------------
int jjs_debug_fq_mul_dev(const void* a, const void* b, size_t n, void* out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!all_ok(a, b, out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dbg_fq_mul_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint8_t*)a, (const uint8_t*)b, (uint64_t)n, (uint8_t*)out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
int jjs_debug_poseidon_dev(const void* in, size_t k, size_t n, void* out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (k < 1 || k > JJS_MAX_HASH_INPUTS) return fail(JJS_ERR_ARG, "k out of range");
    if (n == 0) return JJS_OK;
    if (!all_ok(in, out)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dbg_poseidon_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint8_t*)in, (uint32_t)k, (uint64_t)n, (uint8_t*)out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
int jjs_debug_point_flags_dev(const void* points, size_t n, void* out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!points || !aligned16(points) || !out) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dbg_point_flags_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s,
                       (const uint8_t*)points, (uint64_t)n, (uint8_t*)out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
int jjs_debug_half_scalars_dev(const void* c, size_t n, void* a_out, void* b_out, void* b_neg_out, void* stream) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (n == 0) return JJS_OK;
    if (!all_ok(c, a_out, b_out) || !b_neg_out) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipLaunchKernelGGL(dbg_half_scalars_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t*)c, (uint64_t)n, (uint8_t*)a_out, (uint8_t*)b_out, (uint8_t*)b_neg_out);
    HIP_TRY(hipGetLastError());
    return JJS_OK;
}
#if defined(JJS_PROFILING)
// include/jjs_gpu_profiling.h: these two exist only in libjjs_gpu_prof.so
int jjs_debug_skip_phases(unsigned mask) {
    std::lock_guard<std::mutex> lock(L.mu);
    g_skip_phases = mask & 31u;
    return JJS_OK;
}
int jjs_debug_force_path(int which) {
    std::lock_guard<std::mutex> lock(L.mu);
    g_force_path = which & 3;                       // 0 by size and keys, 1 throughput (key tables allowed), 2 latency, 3 throughput without key tables
    g_force_positions = ((which >> 4) & 15) == 4 || ((which >> 4) & 15) == 8 ? ((which >> 4) & 15) : (((which >> 4) & 15) == 15 ? 16 : 0);    // 0x42 / 0x82 / 0xF2: latency path, 4 / 8 / 16 pieces
    g_keep_order = (which & 0x1000) != 0;
    g_force_window = ((which >> 8) & 15) == KT_WINDOW_NARROW ? ((which >> 8) & 15) : 0;   // 0x500: narrow key-table windows whatever the keys
    return JJS_OK;
}
int jjs_debug_host_timing(double out[8]) {
    std::lock_guard<std::mutex> lock(L.mu);
    for (int i = 0; i < 8; ++i) out[i] = g_host_timing[i];
    return JJS_OK;
}
int jjs_debug_pin_hash_seed(int on) {
    std::lock_guard<std::mutex> lock(L.mu);
    g_pin_hash_seed = on != 0;
    return JJS_OK;
}
int jjs_debug_fail_key_arena(int on) {
    std::lock_guard<std::mutex> lock(L.mu);
    g_fail_key_arena = on != 0;
    return JJS_OK;
}
int jjs_debug_allow_virtual_devices(int allow) {
    std::lock_guard<std::mutex> lock(L.mu);
    g_allow_virtual = allow != 0;
    return JJS_OK;
}
#endif
// Loads RCCL, forms a one-rank clique on the current device and sums a known 4 x u64 vector in place: checks
// the library, the symbols and the call sequence of allreduce_tallies() on a box with a single GPU.
int jjs_debug_rccl_selftest(void) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (int rc = load_rccl()) return rc;
    ncclComm_t comm;
    int ord = g->device;
    RCCL_TRY(L.rccl.CommInitAll(&comm, 1, &ord));
    const unsigned long long in[4] = {3, 1, 4, 0x100000001ull};
    unsigned long long out[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(g->tally, in, sizeof(in), hipMemcpyHostToDevice, g->stream));
    ncclResult_t r = L.rccl.GroupStart();
    if (r == ncclSuccess) r = L.rccl.AllReduce(g->tally, g->tally, 4, ncclUint64, ncclSum, comm, g->stream);
    ncclResult_t r2 = L.rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    hipError_t e = hipMemcpyAsync(out, g->tally, sizeof(out), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    (void)L.rccl.CommDestroy(comm);
    if (r != ncclSuccess) return fail(JJS_ERR_COLLECTIVE, "RCCL self-test: %s", L.rccl.GetErrorString(r));
    if (e != hipSuccess) return fail(JJS_ERR_HIP, "RCCL self-test: %s", hipGetErrorString(e));
    if (memcmp(in, out, sizeof(in)) != 0) return fail(JJS_ERR_COLLECTIVE, "RCCL self-test: wrong sum");
    return JJS_OK;
}
size_t jjs_debug_comb_table_bytes(void) { return COMB_TABLE_WORDS * sizeof(uint32_t); }
int jjs_debug_comb_table(int which, void* host_out) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    if (!host_out) return fail(JJS_ERR_ARG, "null pointer");
    HIP_TRY(hipMemcpy(host_out, which ? g->comb_gn : g->comb_g, COMB_TABLE_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return JJS_OK;
}

}  // extern "C"
