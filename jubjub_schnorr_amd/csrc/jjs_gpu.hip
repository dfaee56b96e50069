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
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <random>
#include <sched.h>
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
#include "safe_tag.h"
#include "jjs_sponge_tags_long.inc"

using namespace jjs;

namespace {

constexpr int BLOCK = 256;
// hash transcripts are indexed with an int (3 + 4 n inputs): far beyond anything a device lane can hash in one piece
constexpr uint32_t JJS_MSIG_PARTICIPANTS_LIMIT = 1u << 24;

// First kernel of a batch: everything that does not need the window tables (see prepare_item).  No
// per-lane workspace, about half the registers of verify_kernel: four waves per SIMD.
// phase: PREP_ALL, or PREP_HEAD / PREP_TAIL for a batch whose keys are still being counted when the launch starts
// (verify_core.h prep_phase); the tail leaves at once when the key tables engaged.
// The launch covers the items [first, first + count) of the batch: a host-buffer call hashes its items range by range
// while the later ranges are still being uploaded (run_host_block); every other call passes (0, n).
__global__ __launch_bounds__(BLOCK, 4) void prepare_kernel(verify_params P, int phase, uint64_t first, uint64_t count) {
    if (phase == PREP_TAIL && keyed_mode(P)) return;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += total) {
        const uint64_t item = first + i;
        store_prep(P.prep, P.n, item, phase == PREP_TAIL ? prepare_tail(P, item, load_prep(P.prep, P.n, item))
                                                         : prepare_item(P, item, true, -1, (prep_phase)phase));
    }
}

// What a first-pass lane does with its verdict: final statuses go to the caller's array and the tally (wave
// ballots, one atomic per status per wave); undecided items (their points still need their own subgroup tests)
// are appended to the queue of the resolve pass (one atomic per wave, entries of a wave contiguous).
__device__ __forceinline__ void publish_status(const verify_params& P, uint64_t item, bool active, uint32_t st) {
    if (active && st < ST_PENDING_EQ_FAILED && P.status) P.status[item] = (uint8_t)st;
    if (P.tally) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            unsigned long long b = __ballot(active && st == k);
            if ((threadIdx.x & 63) == 0 && b) atomicAdd(&P.tally[k], (unsigned long long)__popcll(b));
        }
    }
    const bool pend = active && st >= ST_PENDING_EQ_FAILED;
    const unsigned long long pmask = __ballot(pend);
    if (pmask) {
        const uint32_t lane = threadIdx.x & 63;
        unsigned long long slot = 0;
        if (lane == 0) slot = atomicAdd(P.pending_count, (unsigned long long)__popcll(pmask));
        slot = __shfl(slot, 0);
        if (pend) P.pending[slot + __popcll(pmask & ((1ull << lane) - 1ull))] = (item << 1) | (st == ST_PENDING_EQ_HELD ? 1u : 0u);
    }
}

// second launch-bound argument: at least 2 waves per SIMD, i.e. at most 256 registers per lane
__global__ __launch_bounds__(BLOCK, 2) void verify_kernel(verify_params P) {
    if (keyed_mode(P)) return;                   // this batch went down the key-table path (key_verify_kernel)
    const uint64_t gtid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    uint32_t* ws = P.workspace + gtid * WS_WORDS_PER_LANE;
    for (uint64_t base = 0; base < P.n; base += total) {
        const uint64_t item = base + gtid;
        const bool active = item < P.n;
        const uint64_t it = active ? item : P.n - 1;
        publish_status(P, item, active, finish_item(P, it, ws, load_prep(P.prep, P.n, it)));
    }
}

// ---- key-table path (key_tables.h) ---------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void key_dedup_kernel(key_params K) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t item = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; item < K.n; item += total) {
        for (uint32_t c = 0; c < K.n_cols; ++c) {
            const key_column C = kt_col(K, (int32_t)c);
            uint32_t slot = (uint32_t)kt_hash(C.src, item, C.key_bytes, K.seed) & C.hash_mask;
            uint32_t rep = (uint32_t)item;
            bool settled = false;
            // every probe either claims a slot or meets a settled one; the table has at least 2 n slots, so honest
            // keys settle within a few probes.  Keys crafted to share a slot do not get to make this loop long: after
            // KT_MAX_PROBES the batch gives up on key tables (counters[3]) and takes the throughput path.
            for (uint32_t probe = 0; probe < KT_MAX_PROBES && !settled; ++probe) {
                // look before claiming: with few distinct keys nearly every lane finds its slot taken, and a million
                // compare-and-swaps on one address would queue up behind each other (a stale zero only costs the swap)
                uint32_t cur = __atomic_load_n(&C.hash[slot], __ATOMIC_RELAXED);
                if (cur == 0u) cur = atomicCAS(&C.hash[slot], 0u, (uint32_t)item + 1u);
                if (cur == 0u) settled = true;
                else if (kt_same_key(C.src, item, cur - 1u, C.key_bytes)) { rep = cur - 1u; settled = true; }
                else slot = (slot + 1u) & C.hash_mask;
            }
            if (!settled) atomicOr(&K.counters[3], 1u);
            C.rep[item] = rep;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void key_assign_kernel(key_params K) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK, first = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t base = 0; base < K.n; base += total) {           // wave-uniform trip count: ballots below
        const uint64_t item = base + first;
        for (uint32_t c = 0; c < K.n_cols; ++c) {
            const key_column C = kt_col(K, (int32_t)c);
            const bool is_rep = item < K.n && C.rep[item] == (uint32_t)item;
            // one atomic per wave (a batch of unique keys would otherwise put 2^20 atomics on one counter)
            const unsigned long long m = __ballot(is_rep);
            if (!m) continue;
            uint32_t start = 0;
            if (lane == (uint32_t)__ffsll((long long)m) - 1u) start = atomicAdd(&K.counters[c], (uint32_t)__popcll(m));
            start = (uint32_t)__shfl((int)start, __ffsll((long long)m) - 1);
            if (is_rep) {
                const uint32_t id = start + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                C.keyid[item] = id;
                if (id < K.max_keys) C.key_item[id] = (uint32_t)item;
            }
        }
    }
}
__global__ __launch_bounds__(BLOCK) void key_spread_kernel(key_params K) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // the decision: enough signatures per key in every column
        bool use = K.counters[3] == 0u, wide = true;       // no probe sequence was cut short
        bool fits_narrow = true;
        for (uint32_t c = 0; c < K.n_cols; ++c) {
            use = use && (uint64_t)K.counters[c] * KT_MIN_MULTIPLICITY <= K.n;
            // wide windows where the keys repeat enough to repay them AND the slot's table pool holds that many wide tables
            wide = wide && (uint64_t)K.counters[c] * KT_WIDE_MULTIPLICITY <= K.n && K.counters[c] <= K.max_keys_wide;
            fits_narrow = fits_narrow && K.counters[c] <= K.max_keys;
        }
        const uint32_t w = (wide && K.force_window != (uint32_t)KT_WINDOW_NARROW) ? KT_WINDOW_WIDE : KT_WINDOW_NARROW;
        // keys that repeat but whose tables do not fit the pool: this batch takes the throughput path, the host reads
        // counters[4] back after the call and the pool has grown by the next one (note_key_feedback)
        if (use && w == (uint32_t)KT_WINDOW_NARROW && !fits_narrow) { use = false; K.counters[4] = 1u; }
        K.counters[2] = use ? w : 0u;                      // ... and the window width of the tables
    }
    for (uint64_t item = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; item < K.n; item += total)
        for (uint32_t c = 0; c < K.n_cols; ++c) {
            const key_column C = kt_col(K, (int32_t)c);
            const uint32_t r = C.rep[item];
            if (r != (uint32_t)item) C.keyid[item] = C.keyid[r];      // r's own id was written by the previous launch
        }
}
__global__ __launch_bounds__(BLOCK, 2) void key_chain_kernel(key_params K) {
    const int w = (int)K.counters[2];
    if (!w) return;
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x, c = t / K.max_keys, id = t % K.max_keys;
    if (c < K.n_cols && id < K.counters[c]) kt_chain_key(kt_col(K, (int32_t)c), id, w);
}
// the grid covers max_keys x KT_MAX_POSITIONS lanes per column; a batch with wide windows has fewer of both
__global__ __launch_bounds__(BLOCK, 2) void key_table_kernel(key_params K) {
    const int w = (int)K.counters[2];
    if (!w) return;
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t positions = (uint32_t)kt_positions(w);
    const uint64_t per_col = (uint64_t)K.max_keys * positions;
    const uint32_t c = (uint32_t)(t / per_col), id = (uint32_t)((t % per_col) / positions), pos = (uint32_t)(t % positions);
    if (c < K.n_cols && id < K.counters[c]) kt_table_lane(kt_col(K, (int32_t)c), id, pos, w);
}
// Items grouped by key (column 0): histogram, exclusive scan, scatter.  All three leave at once when the batch does not
// take the key-table path.
// counters[key] += 1 for every active lane; returns the lane's slot (the counter before the addition, plus the lane's
// rank among the lanes that were added together).  Lanes that share a key with many others of the wave are added with
// one atomic per key (up to WAVE_GROUPS keys per wave): a batch under a handful of keys would otherwise put 2^20 atomics
// on a handful of addresses (measured: 2 keys, 20.8 ms a batch instead of 9).  The other lanes add one by one: a wave
// with many distinct keys has no contention to avoid, and a turn of the grouping loop per key would cost it more
// (measured: 64 turns, +0.8 ms a batch), so the loop stops at the first key that is rare in the wave.  Every lane of
// the wave must call it (ballots and shuffles).
constexpr int WAVE_GROUPS = 8;
// A batch under a few keys keeps the cursor of key k at key_cursor[k * CURSOR_STRIDE], a 64-byte line each, so that its
// atomics do not all land on one line and one L2 channel, which the hashes running beside them also need (16 keys: 12.2 ->
// 10.2 ms a batch); from CURSOR_DENSE_FROM keys on the cursors are dense (padded ones cost the SURVEY workload 2.7 %).
constexpr uint32_t CURSOR_STRIDE = 16, CURSOR_DENSE_FROM = 65;
__device__ __forceinline__ uint32_t cursor_stride(const key_params& K) { return K.counters[0] < CURSOR_DENSE_FROM ? CURSOR_STRIDE : 1u; }
__device__ __forceinline__ uint32_t wave_grouped_add(uint32_t* counters, uint32_t stride, uint32_t key, bool active) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t leader = lane, rank = 0, size = 1;
    unsigned long long todo = __ballot(active);
#pragma unroll 1
    for (int turn = 0; turn < WAVE_GROUPS && todo; ++turn) {          // wave-uniform; no memory access in here
        const int first = __ffsll((long long)todo) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, first);
        const unsigned long long same = __ballot(active && key == k) & todo;
        if ((same >> lane) & 1ull) {
            leader = (uint32_t)first;
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            size = (uint32_t)__popcll(same);
        }
        todo &= ~same;
        if (__popcll(same) < 4) break;                                // a rare key: the wave is not one of few keys
    }
    // all the atomics of the wave in one go: a group's leader for its group, every ungrouped lane for itself
    uint32_t base = 0;
    if (active && lane == leader) base = atomicAdd(&counters[(size_t)key * stride], size);
    base = (uint32_t)__shfl((int)base, (int)leader);
    return base + rank;
}
__global__ __launch_bounds__(BLOCK) void key_count_kernel(key_params K) {
    if (!K.counters[2]) return;
    const uint32_t stride = cursor_stride(K);
    const uint64_t total = (uint64_t)gridDim.x * BLOCK, first = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (uint64_t base = 0; base < K.n; base += total) {              // wave-uniform trip count
        const uint64_t item = base + first;
        const bool active = item < K.n;
        (void)wave_grouped_add(K.key_cursor, stride, active ? K.col[0].keyid[item] : 0u, active);
    }
}
__global__ __launch_bounds__(1024) void key_scan_kernel(key_params K) {          // one block
    if (!K.counters[2]) return;
    const uint32_t stride = cursor_stride(K);
    __shared__ uint32_t part[1024];
    const uint32_t keys = K.counters[0], per = (keys + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < keys ? lo + per : keys;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += K.key_cursor[(size_t)k * stride];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;                 // exclusive prefix of this thread's keys
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t c = K.key_cursor[(size_t)k * stride]; K.key_cursor[(size_t)k * stride] = run; run += c; }
}
__global__ __launch_bounds__(BLOCK) void key_scatter_kernel(key_params K) {
    if (!K.counters[2]) return;
    const uint32_t stride = cursor_stride(K);
    const uint64_t total = (uint64_t)gridDim.x * BLOCK, first = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (uint64_t base = 0; base < K.n; base += total) {              // wave-uniform trip count
        const uint64_t item = base + first;
        const bool active = item < K.n;
        const uint32_t slot = wave_grouped_add(K.key_cursor, stride, active ? K.col[0].keyid[item] : 0u, active && !K.keep_order);
        if (active) K.order[K.keep_order ? (uint32_t)item : slot] = (uint32_t)item;
    }
}
__global__ __launch_bounds__(BLOCK, 2) void key_verify_kernel(verify_params P, key_params K) {
    if (!keyed_mode(P)) return;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t base = 0; base < P.n; base += total) {
        const uint64_t idx = base + (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
        const bool active = idx < P.n;
        const uint64_t item = K.order[active ? idx : P.n - 1];
        publish_status(P, item, active, kt_finish_item(P, K, item, load_prep(P.prep, P.n, item)));
    }
}

// Second pass: the queued items, densely packed over the lanes, P.resolve_lanes adjacent lanes per item
// (one point each; see verify_item / resolve_point).
__global__ __launch_bounds__(BLOCK) void resolve_kernel(verify_params P) {
    const uint64_t gtid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    const uint64_t count = *P.pending_count;
    const uint32_t L = P.resolve_lanes;
    for (uint64_t base = 0; base < count * L; base += total) {
        const uint64_t slot = base + gtid;
        const uint64_t idx = slot / L;
        const uint32_t j = (uint32_t)(slot % L);
        const bool active = idx < count;
        if (!__ballot(active)) break;                       // a wave past the end of the queue has nothing to do
        const uint64_t e = P.pending[active ? idx : count - 1];
        const uint64_t item = e >> 1;
        bool tf = resolve_point(P, item, j);
        for (uint32_t d = 1; d < L; d <<= 1) tf = (__shfl_xor((int)tf, (int)d) != 0) && tf;
        const uint32_t st = resolve_status(tf, (e & 1u) != 0);
        const bool writer = active && j == 0;
        if (writer && P.status) P.status[item] = (uint8_t)st;
        if (P.tally) {
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                unsigned long long b = __ballot(writer && st == k);
                if ((threadIdx.x & 63) == 0 && b) atomicAdd(&P.tally[k], (unsigned long long)__popcll(b));
            }
        }
    }
}

// ---- latency path for small batches (small_batch.h) ------------------------------------------------------
// Phase A: the three roles share one launch; the role of a block follows from its index, the longest-running
// blocks first (hash, then the chains from the far position down, then the point checks).
__global__ __launch_bounds__(BLOCK, 2) void small_a_kernel(small_params S, uint32_t hash_blocks, uint32_t chain_blocks_per_pos) {
    const uint32_t b = blockIdx.x;
    const uint64_t n = S.V.n;
    if (b < hash_blocks) {
        __builtin_amdgcn_s_setprio(3);            // the critical path: ahead of co-resident chain / point waves
        const uint64_t idx = (uint64_t)b * BLOCK + threadIdx.x;
        if (S.hash_lanes == 1) {
            if (idx < n) sb_hash_item(S, idx);
            return;
        }
        const uint64_t item = idx / SB_HASH_LANES;
        const bool active = item < n;             // whole groups of eight lanes: the shuffles of a group stay inside it
        sb_hash_item_coop(S, active ? item : n - 1, (int)(idx % SB_HASH_LANES), active);
        return;
    }
    const uint32_t cb = b - hash_blocks;
    if (cb < S.positions * chain_blocks_per_pos) {
        const uint32_t k = S.positions - 1 - cb / chain_blocks_per_pos;            // block-uniform position
        const uint64_t r = (uint64_t)(cb % chain_blocks_per_pos) * BLOCK + threadIdx.x;
        const uint32_t per_item = 2 * S.V.n_eq;                                     // (equation, PK | R)
        if (r < n * per_item) sb_chain_lane(S, r / per_item, (uint32_t)(r % per_item) >> 1, (uint32_t)r & 1u, k);
        return;
    }
    const uint64_t r = (uint64_t)(cb - S.positions * chain_blocks_per_pos) * BLOCK + threadIdx.x;
    if (r < n * S.V.n_points) sb_point_lane(S, r / S.V.n_points, (uint32_t)(r % S.V.n_points));
}

// lane ^ 1 and lane ^ 2 inside a group of four lanes: one DPP move per word
template <int CTRL>
__device__ __forceinline__ ext_pt dpp_quad(const ext_pt& p) {
    ext_pt r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.x.l[i], CTRL, 0xf, 0xf, false);
        r.y.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.y.l[i], CTRL, 0xf, 0xf, false);
        r.z.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.z.l[i], CTRL, 0xf, 0xf, false);
        r.t.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p.t.l[i], CTRL, 0xf, 0xf, false);
    }
    return r;
}
__device__ __forceinline__ ext_pt shfl_xor_ext(const ext_pt& p, int mask) {
    ext_pt r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = (uint32_t)__shfl_xor((int)p.x.l[i], mask);
        r.y.l[i] = (uint32_t)__shfl_xor((int)p.y.l[i], mask);
        r.z.l[i] = (uint32_t)__shfl_xor((int)p.z.l[i], mask);
        r.t.l[i] = (uint32_t)__shfl_xor((int)p.t.l[i], mask);
    }
    return r;
}
// Phase B: `positions` adjacent lanes per equation (twice that per item for the double scheme); every lane of a
// group ends with the whole left side of its equation, lane 0 of the item writes the verdict.
__global__ __launch_bounds__(BLOCK, 2) void small_b_kernel(small_params S) {
    const uint64_t n = S.V.n;
    const uint32_t pos = S.positions, lanes_per_item = pos * S.V.n_eq;
    const uint64_t total = n * lanes_per_item;
    const uint64_t idx = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = idx < total;              // groups are whole: total is a multiple of the group size
    const uint64_t ii = active ? idx : total - 1;
    const uint64_t item = ii / lanes_per_item;
    const uint32_t sub = (uint32_t)(ii % lanes_per_item), e = sub / pos, k = sub % pos;
    const prep_record r = load_prep(S.V.prep, n, item);
    ext_pt acc = sb_piece(S, item, e, k, r);
    acc = sb_add(acc, dpp_quad<0xB1>(acc));       // quad_perm [1,0,3,2]: partner lane ^ 1
    acc = sb_add(acc, dpp_quad<0x4E>(acc));       // quad_perm [2,3,0,1]: partner lane ^ 2
    if (pos == 8) acc = sb_add(acc, shfl_xor_ext(acc, 4));
    bool eq_ok = sb_equation_holds(S, item, e, acc);
    if (S.V.n_eq == 2) eq_ok = (__shfl_xor((int)eq_ok, (int)pos) != 0) && eq_ok;
    const uint32_t st = sb_status(r.malformed, sb_points_ok(S, item), eq_ok);
    const bool writer = active && sub == 0;
    if (writer && S.V.status) S.V.status[item] = (uint8_t)st;
    if (S.V.tally) {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            unsigned long long bal = __ballot(writer && st == c);
            if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&S.V.tally[c], (unsigned long long)__popcll(bal));
        }
    }
}

struct challenge_params {
    uint32_t n_hash, pad_;
    fe_src hash_in[10];
    uint64_t n;
    uint8_t* c_out;
};
__global__ __launch_bounds__(BLOCK) void challenge_kernel(challenge_params P) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t item = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; item < P.n; item += total) {
        fe_n d = poseidon_digest((int)P.n_hash, [&](int e) { return load_fq(P.hash_in[e], item); });
        store_words(P.c_out, item, truncate250(d));
    }
}

struct decode_params {
    uint32_t n_src, pad_;
    fe_src src[4];        // compressed points: 32 bytes at base + i*stride + off
    uint8_t* out[4];      // affine u || v, n x 64 each
    uint8_t* bad;         // n bytes, set to 1 when any source of item i fails to decode (nullable)
    uint8_t* ok;          // n bytes, 1/0 per item for source 0 (nullable; debug entry point)
    uint64_t n;           // items of this launch: first .. first + n - 1
    uint64_t first;
    dlog_tables dlog;
    const uint32_t* skip_flag;   // nullable: the launch leaves at once when the word is non-zero (keys decoded per key instead)
};
__global__ __launch_bounds__(BLOCK) void dlog_table_kernel(uint32_t* pow, uint8_t* hash) {
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t < 7 * 256) dlog_table_entry(pow, hash, t / 256, t % 256);
}
__global__ __launch_bounds__(BLOCK) void decode_kernel(decode_params P) {
    if (P.skip_flag && *P.skip_flag) return;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < P.n; i += total) {
        const uint64_t item = P.first + i;
        bool all_ok = true;
        for (uint32_t k = 0; k < P.n_src; ++k) {
            decoded_point d = decompress_point(load_words(P.src[k], item), P.dlog);
            store_words(P.out[k], 2 * item, d.u);
            store_words(P.out[k], 2 * item + 1, d.v);
            all_ok = all_ok && d.ok;
        }
        if (P.bad && !all_ok) P.bad[item] = 1;
        if (P.ok) P.ok[item] = all_ok ? 1 : 0;
    }
}
// Wire calls on the key-table path: one decompression per distinct key, then every item copies its key's point
// (kt_decode_key / kt_unpack_item); both leave at once when the batch does not take the key-table path.
struct key_decode_params {
    uint8_t* out[2];      // decoded affine column (n x 64) of key column 0 / 1
    uint8_t* bad;         // per-item malformed flags of the call
    dlog_tables dlog;
};
__global__ __launch_bounds__(BLOCK) void key_decode_kernel(key_params K, key_decode_params D) {
    if (!K.counters[2]) return;
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x, c = t / K.max_keys, id = t % K.max_keys;
    if (c < K.n_cols && id < K.counters[c]) kt_decode_key(kt_col(K, (int32_t)c), id, c == 0 ? D.out[0] : D.out[1], D.dlog);
}
__global__ __launch_bounds__(BLOCK) void key_unpack_kernel(key_params K, key_decode_params D, uint64_t first, uint64_t count) {
    if (!K.counters[2]) return;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += total)
        for (uint32_t c = 0; c < K.n_cols; ++c) kt_unpack_item(kt_col(K, (int32_t)c), first + i, c == 0 ? D.out[0] : D.out[1], D.bad);
}
// (U, V, Z) -> affine for the *_ext entry points: every lane owns the items lane, lane + lanes, ... and shares one
// field inversion among them (normalize.h)
__global__ __launch_bounds__(BLOCK) void normalize_kernel(normalize_params P) {
    __builtin_amdgcn_s_setprio(3);            // few waves, a long dependent chain, and the hashes of their items wait for them
    normalize_lane(P, (uint64_t)blockIdx.x * BLOCK + threadIdx.x, (uint64_t)gridDim.x * BLOCK);
}
__global__ __launch_bounds__(BLOCK) void compress_kernel(const uint8_t* affine, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_src s{affine, 64, 0};
    store_words(out, i, compress_point(load_words(s, i), load_words(s, i, 32)));
}

__global__ __launch_bounds__(BLOCK) void sign_kernel(sign_params P) {
    const uint64_t gtid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    uint32_t* ws = P.workspace + gtid * WS_WORDS_PER_LANE;
    for (uint64_t item = gtid; item < P.n; item += total) sign_item(P, item, ws);
}

// PublicKey::from(&SecretKey) = sk * G (reference src/keys/public.rs:54-60) and the second half of
// PublicKeyDouble::from (sk * G', src/keys/public/double.rs:47-57): fixed-base only, NOT constant time.
__global__ __launch_bounds__(BLOCK) void derive_kernel(const uint8_t* sk, uint64_t n, const uint32_t* comb_g,
                                                       const uint32_t* comb_gn, uint8_t* pk_out, uint8_t* pkp_out,
                                                       uint8_t* bad) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    const fe_src s_sk{sk, 32, 0};
    for (uint64_t item = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; item < n; item += total) {
        const words8 k = load_words(s_sk, item);
        if (bad) bad[item] = words_lt(k, JJS_FR_WORDS) ? 0 : 1;       // non-canonical scalar (>= r)
        store_point(pk_out, item, to_affine_words(comb_mul(comb_g, k)));
        if (pkp_out) store_point(pkp_out, item, to_affine_words(comb_mul(comb_gn, k)));
    }
}

// multisig passes: 0 map, 1 delinearisation, 2 aggregate key + a, 3 commitments, 4 RSa + c + u, 5 shares, 6 verdicts
__global__ __launch_bounds__(BLOCK) void msig_kernel(msig_params P, int pass) {
    const uint64_t gtid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    uint32_t* ws = P.lane_ws + gtid * WS_WORDS_PER_LANE;
    const bool per_transcript = (pass == 0 || pass == 2 || pass == 4 || pass == 6);
    const uint64_t count = per_transcript ? P.n_transcripts : P.n_total;
    for (uint64_t i = gtid; i < count; i += total) {
        switch (pass) {
        case 0: msig_map_item(P, (uint32_t)i); break;
        case 1: msig_delin_item(P, i, ws); break;
        case 2: msig_agg_item(P, (uint32_t)i); break;
        case 3: msig_commit_item(P, i, ws); break;
        case 4: msig_final_item(P, (uint32_t)i); break;
        case 5: msig_share_item(P, i, ws); break;
        default: msig_verdict_item(P, (uint32_t)i); break;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void comb_kernel(uint32_t* table, int which) {
    int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= COMB_WINDOWS * COMB_ENTRIES) return;
    build_comb_entry(table, which ? JJS_GN : JJS_G, t / COMB_ENTRIES, t % COMB_ENTRIES);
}

__global__ __launch_bounds__(BLOCK) void dbg_fq_mul_kernel(const uint8_t* a, const uint8_t* b, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_src sa{a, 32, 0}, sb{b, 32, 0};
    store_words(out, i, fq_to_words(fq_mul(load_fq(sa, i), load_fq(sb, i))));
}
__global__ __launch_bounds__(BLOCK) void dbg_poseidon_kernel(const uint8_t* in, uint32_t k, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_src s{in, 32 * k, 0};
    fe_n d = poseidon_digest((int)k, [&](int e) { return load_fq(s, i, 32u * (uint32_t)e); });
    store_words(out, i, fq_to_words(d));
}
__global__ __launch_bounds__(BLOCK) void dbg_point_flags_kernel(const uint8_t* pts, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    fe_src s{pts, 64, 0};
    fe_n u = load_fq(s, i), v = load_fq(s, i, 32);
    bool id = affine_is_identity(u, v);
    out[i] = (uint8_t)((affine_on_curve(u, v) ? 1 : 0) | ((id || is_torsion_free(u, v)) ? 2 : 0) | (id ? 4 : 0) |
                       (is_torsion_free_by_order(u, v) ? 8 : 0));
}

// half_size_scalars as the device runs it (v_rcp_f64 estimates, wave ballots for loop control), so that the
// adversarial inputs of tests/test_hostbuild.py reach the GPU code path too.  Every lane of a wave runs the
// loop (the last item is repeated in the tail).
__global__ __launch_bounds__(BLOCK) void dbg_half_scalars_kernel(const uint8_t* c, uint64_t n, uint8_t* a_out, uint8_t* b_out,
                                                                 uint8_t* neg_out) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n;
    const fe_src s{c, 32, 0};
    const half_scalars h = half_size_scalars(load_words(s, active ? i : n - 1));
    if (!active) return;
    reinterpret_cast<u32x4*>(a_out)[i] = u32x4{h.a.w[0], h.a.w[1], h.a.w[2], h.a.w[3]};
    reinterpret_cast<u32x4*>(b_out)[i] = u32x4{h.b.w[0], h.b.w[1], h.b.w[2], h.b.w[3]};
    neg_out[i] = h.b_neg ? 1 : 0;
}

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
    uint64_t seen_n = 0;              // ... whose batch had this many items in
    uint32_t seen_cols = 0;           // ... this many key columns
    hipStream_t key_stream = nullptr; // the per-key kernels of the slot's call run here, beside the challenge hashes
    hipEvent_t key_fork = nullptr, key_mid = nullptr, key_join = nullptr;
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

constexpr size_t HOST_MAX_PIECES = 40;      // pieces a host-buffer call uploads its block in (plan_pieces)
#ifndef JJS_HOST_SIDE_STREAMS
#define JJS_HOST_SIDE_STREAMS 3
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
// ---- key-table path: arenas ------------------------------------------------------------------------------
// Batches of at least this many items (big or medium slot) try the key tables.
constexpr size_t KT_MIN_ITEMS = 65536;
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
    if (sl->keys) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(sl->keys));
        sl->keys = nullptr; sl->keys_bytes = 0;
    }
    HIP_TRY(hipMalloc(&sl->keys, bytes));
    sl->keys_bytes = bytes;
    return JJS_OK;
}
// The pool grows when a call has asked for more; when hipMalloc says no, the pool the slot has stays (and that size
// is not asked for again).  Nothing is freed before its replacement exists.
int ensure_key_pool() {
#if defined(JJS_PROFILING)
    if (g_fail_key_arena) return fail(JJS_ERR_HIP, "key arena allocation failed (jjs_debug_fail_key_arena)");
#endif
    size_t want = sl->key_pool_want > KEY_POOL_INITIAL_BYTES ? sl->key_pool_want : KEY_POOL_INITIAL_BYTES;
    const bool refused = sl->key_pool_refused && want >= sl->key_pool_refused;     // hipMalloc has said no to this much before
    if (sl->key_pool && (want <= sl->key_pool_bytes || refused)) return JJS_OK;
    if (!sl->key_pool && refused) want = KEY_POOL_INITIAL_BYTES;
    uint8_t* fresh = nullptr;
    if (hipMalloc(&fresh, want) != hipSuccess) {
        (void)hipGetLastError();
        sl->key_pool_refused = want;
        ++g->stats[JJS_PATH_KEYS_NO_MEMORY];
        return sl->key_pool ? JJS_OK : fail(JJS_ERR_HIP, "hipMalloc of the key-table pool (%zu bytes) failed", want);
    }
    if (sl->key_pool) {
        HIP_TRY(hipDeviceSynchronize());        // earlier launches may still read the old pool
        HIP_TRY(hipFree(sl->key_pool));
    }
    sl->key_pool = fresh;
    sl->key_pool_bytes = want;
    return JJS_OK;
}

bool key_path_applies(const verify_params& P) {
#if defined(JJS_PROFILING)
    if (g_force_path == 3) return false;           // throughput path without the key tables
#endif
    return P.n >= KT_MIN_ITEMS && P.n <= 0x7fffffffu && P.n_eq >= 1 && sl->key_stream != nullptr;
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
    static std::mt19937_64 rng = [] {
        std::random_device rd;
        std::seed_seq seq{rd(), rd(), rd(), rd(), (unsigned)std::chrono::steady_clock::now().time_since_epoch().count()};
        return std::mt19937_64(seq);
    }();
    return rng();
}
// Carves the key buffers of this call out of the slot's two arenas and clears the hash tables and counters (on `s`).
int setup_keys(const verify_params& P, key_params& K, hipStream_t s) {
    note_key_feedback();
    K.n = P.n;
    K.seed = next_seed();
#if defined(JJS_PROFILING)
    K.force_window = (uint32_t)g_force_window;
    K.keep_order = g_keep_order ? 1u : 0u;
    if (g_pin_hash_seed) K.seed = 0;
#endif
    // key columns: PK of every equation, and the generator where it is per-item data
    fe_src cols[2];
    uint32_t n_cols = 0;
    for (uint32_t e = 0; e < P.n_eq; ++e) {
        cols[P.eq[e].pk_col] = P.eq[e].pk; n_cols = n_cols > (uint32_t)P.eq[e].pk_col + 1 ? n_cols : (uint32_t)P.eq[e].pk_col + 1;
        if (!P.eq[e].comb) { cols[P.eq[e].gen_col] = P.eq[e].gen; n_cols = n_cols > (uint32_t)P.eq[e].gen_col + 1 ? n_cols : (uint32_t)P.eq[e].gen_col + 1; }
    }
    K.n_cols = n_cols;
    if (int rc = ensure_key_pool()) return rc;
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
    if (int rc = ensure_key_index(256 + order_bytes + n_cols * per_col)) return rc;
    uint8_t* p = sl->keys;
    K.counters = reinterpret_cast<uint32_t*>(p); p += 256;
    HIP_TRY(hipMemsetAsync(K.counters, 0, 256, s));
    K.order = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
    K.key_cursor = reinterpret_cast<uint32_t*>(p); p += pad256(cursor_words * 4);
    HIP_TRY(hipMemsetAsync(K.key_cursor, 0, cursor_words * 4, s));
    for (uint32_t c = 0; c < n_cols; ++c) {
        key_column& C = K.col[c];
        C.src = cols[c];
        C.key_bytes = 64;
        C.hash = reinterpret_cast<uint32_t*>(p); C.hash_mask = (uint32_t)(slots - 1); p += pad256(slots * 4);
        HIP_TRY(hipMemsetAsync(C.hash, 0, slots * 4, s));
        C.rep = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
        C.keyid = reinterpret_cast<uint32_t*>(p); p += pad256(P.n * 4);
        uint8_t* q = sl->key_pool + (size_t)c * col_bytes;
        C.key_item = reinterpret_cast<uint32_t*>(q); q += L.item_bytes;
        C.key_flags = q; q += L.flag_bytes;
        C.key_undecodable = q; q += L.flag_bytes;
        C.bases = reinterpret_cast<uint32_t*>(q); q += L.base_bytes;
        C.tables = reinterpret_cast<uint32_t*>(q);
    }
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
};

int launch_normalize(normalize_params N, uint64_t first, uint64_t count, uint64_t n_call, hipStream_t s) {
    if (!N.n_src || !count) return JJS_OK;
    N.first = first; N.n = count;
    size_t blocks = (count + BLOCK - 1) / BLOCK;
    const size_t by_share = (count + (size_t)BLOCK * 32 - 1) / ((size_t)BLOCK * 32);
    if (count == n_call) {
        // a whole call: ~8 items per lane at BASELINE sizes (one inversion amortised over them), one item per lane for small calls
        if (blocks > 512) blocks = 512;
    } else {
        // a range of a host-buffer call: ~8 items per lane from 2^18 items on (an inversion is 12 items' worth of products)
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
    if (P.tally) HIP_TRY(hipMemsetAsync(P.tally, 0, 4 * sizeof(unsigned long long), s));
    if (P.pre_malformed) HIP_TRY(hipMemsetAsync(const_cast<uint8_t*>(P.pre_malformed), 0, P.n, s));
    J.small = small_path_applies(P);
    if (J.small) { ++g->stats[JJS_PATH_LATENCY]; return JJS_OK; }
    if (int rc = ensure_pending(P.n)) return rc;
    P.pending_count = reinterpret_cast<unsigned long long*>(sl->pending);
    P.pending = sl->pending + 2;
    HIP_TRY(hipMemsetAsync(sl->pending, 0, sizeof(uint64_t), s));
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
    if (J.C.wire && (cols & COLS_REST)) {              // R (R') of every item
        decode_params D = J.C.W.sig;
        D.first = first; D.n = count;
        D.dlog = dlog_tables{g->dlog_pow, g->dlog_hash};
        hipLaunchKernelGGL(decode_kernel, dim3((unsigned)grid_for(8192, count)), dim3(BLOCK), 0, cs, D);
        HIP_TRY(hipGetLastError());
    }
    return JJS_OK;
}

// Every key column of the call is in place (on the key stream's timeline: the caller has made it wait for whatever
// put them there): count the distinct keys, decide on the device, build the per-key tables.
int job_keys(verify_job& J) {
    if (!J.try_keys) return JJS_OK;
    const verify_params& P = J.C.P;
    key_params& K = J.K;
    hipStream_t ks = sl->key_stream;
    const unsigned item_blocks = (unsigned)grid_for(8192, P.n);
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
    hipLaunchKernelGGL(key_chain_kernel, dim3(key_blocks), dim3(BLOCK), 0, ks, K);
    hipLaunchKernelGGL(key_table_kernel, dim3((unsigned)(((uint64_t)K.n_cols * K.max_keys * KT_MAX_POSITIONS + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, ks, K);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl->key_join, ks));
    J.keys_queued = true;
    return JJS_OK;
}
// a wire call that tries the key tables can hash only after job_keys (it waits for the decoded keys)
bool job_hash_needs_keys(const verify_job& J) { return J.C.wire && J.try_keys; }

// The items [first, first + count) have all their columns in place and ingested: hash them.
int job_hash(verify_job& J, uint64_t first, uint64_t count, hipStream_t cs) {
    if (!count) return JJS_OK;
    const verify_params& P = J.C.P;
    if (J.small) {
        if (first != 0 || count != P.n) return fail(JJS_ERR_ARG, "internal: the latency path takes the call whole");
        if (J.C.wire)
            if (int rc = launch_key_decode_per_item(J, 0, P.n, nullptr, cs)) return rc;
        return launch_small(P, cs);
    }
    if (J.C.wire) {
        if (J.try_keys) {
            if (!J.keys_queued) return fail(JJS_ERR_ARG, "internal: wire hashes before the key kernels");
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
    if (J.forked && hipEventRecord(sl->key_join, sl->key_stream) == hipSuccess) (void)hipStreamWaitEvent(J.s, sl->key_join, 0);
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
    if (!rc) rc = job_hash(J, 0, C.P.n, s);
    if (!rc) rc = job_finish(J);
    if (rc) job_abandon(J);
    return rc;
}
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
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c.seen), sizeof(key_feedback), hipHostMallocDefault));
            memset(c.seen, 0, sizeof(key_feedback));
        }
    }
    // The priority streams, in the order of their importance: the runtime hands its high-priority hardware queues out in
    // creation order, and the key streams of the big slots must have one each (scripts/timeline.sh: created behind two
    // other priority streams, the small launches of the first big slot's key stream waited 0.3-0.5 ms each for wave slots
    // instead of 0.1, and a resident 2^20 batch took 0.5 ms longer).
    {
        const int order[] = {0, SECOND_BIG_SLOT, 1 + N_SMALL_SLOTS, 2 + N_SMALL_SLOTS, 3 + N_SMALL_SLOTS};
        static_assert(N_MEDIUM_SLOTS == 3 && N_BIG_SLOTS == 2, "one entry per slot that has a key stream");
        for (int i : order) HIP_TRY(hipStreamCreateWithPriority(&d.slots[i].key_stream, hipStreamNonBlocking, d.key_priority));
    }
    for (hipStream_t& is : d.ingest) HIP_TRY(hipStreamCreateWithPriority(&is, hipStreamNonBlocking, d.key_priority));
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
        void* sb[] = {c.workspace, c.pending, c.prep, c.wire, c.small, c.keys, c.key_pool};
        for (void* b : sb)
            if (b) (void)hipFree(b);
        if (c.seen) (void)hipHostFree(c.seen);
        hipEvent_t evs[] = {c.last_use, c.key_fork, c.key_mid, c.key_join};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
    }
    if (d.pinned) (void)hipHostFree(d.pinned);
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
    d = device_state{};
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

// Host-buffer calls.  The batch is cut into one contiguous block of ceil(n / devices) items per driven
// device (the rule of jubjub_schnorr_amd/sharding.py) and every block is driven by its OWN host thread, so that
// the uploads of different devices overlap (one thread issuing pageable copies for all devices would stage them
// one after the other).  A block is ONE verification call on its device (verify_job), fed piece by piece: the thread
// copies a piece of the caller's (pageable) arrays into one of two pinned staging slots -- with the help of the device's
// staging threads, a single one moves ~11 GB/s -- queues its upload on the device's copy stream and, behind the upload,
// whatever the piece makes possible: format conversion of its columns, the key kernels once every key has arrived, the
// challenge hashes of the items whose columns are now complete.  The pieces of a block that may take the key tables come
// in this order: all columns of a first few items (so that the hashes start at once), then the KEY columns of all the
// others (the keys of the whole call are counted and tabled once, beside the hashes), then the remaining columns in
// growing ranges.  The equations run once at the end, over the whole block, as in a resident call.  Device arena, pinned
// staging and events are per device and only grow.  The tallies are summed over the devices with one RCCL all-reduce.
// A failing block drains its streams before it reports, so nothing is in flight into the caller's or the library's
// buffers when the call returns an error.
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
    if (g->stage) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(g->stage));
        g->stage = nullptr; g->stage_bytes = 0;
    }
    HIP_TRY(hipMalloc(&g->stage, bytes));
    g->stage_bytes = bytes;
    return JJS_OK;
}
int ensure_pinned(size_t bytes) {
    if (bytes <= g->pinned_bytes) return JJS_OK;
    if (g->pinned) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipHostFree(g->pinned));
        g->pinned = nullptr; g->pinned_bytes = 0;
    }
    HIP_TRY(hipHostMalloc(&g->pinned, bytes, hipHostMallocDefault));
    g->pinned_bytes = bytes;
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
//   * a wire call hashes nothing before its keys are decoded (keys_gate_hashes): its key column goes first, whole.
//   * the columns nothing reads before the equations (u) travel last, behind everything the hashes need, when the call
//     hashes with the head launch (`late`: 16 % fewer bytes ahead of the first hashes of a single batch); else with the others.
// row_keys / row_rest / row_late: bytes per item of the column groups.
void plan_pieces(std::vector<host_piece>& pieces, size_t& largest_bytes, size_t nl, size_t row_keys, size_t row_rest, size_t row_late,
                 bool keys_gate_hashes, bool late) {
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
        size_t lead = 0;
        if (!keys_gate_hashes) {
            size_t next = HOST_LEAD_ITEMS;
            do {
                add(lead, next, COLS_KEYS | rest);
                lead += next;
                next = next * 2 < cap_rest ? next * 2 : cap_rest;
            } while (lead + next <= nl / HOST_LEAD_SHARE_DEN * HOST_LEAD_SHARE_NUM);
        }
        ranges(lead, cap_keys, 1, cap_keys, COLS_KEYS);
        ranges(lead, HOST_REST_ITEMS_FIRST, HOST_REST_GROWTH, cap_rest, rest);
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
    if (int rc = build(in, nl, st, g->tally, g->stream, J.C)) return rc;
    bool late = false;
    for (size_t k = 0; k < n_cols; ++k) late = late || cols[k].group == COLS_LATE;
    late = late && !J.C.wire && !small_path_applies(J.C.P) && key_path_applies(J.C.P) && ensure_key_pool() == JJS_OK;
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
    if (int rc = job_begin(J, g->stream)) return rc;
    // The ranges of the block are queued on several streams in turn: a launch waits for every block of its predecessor on
    // the same stream, and a block of hashes lives for 1.4 ms, so on one stream (or two: scripts/host_timeline.sh) the chip
    // runs half empty at the end of every range; with a stream per range in flight, whichever range has arrived fills the
    // wave slots that come free.  The other streams start behind this one's job_begin (cleared flags and counters).
    HIP_TRY(hipEventRecord(g->host_begin, g->stream));
    hipStream_t compute[1 + HOST_SIDE_STREAMS] = {g->stream};
    for (int k = 0; k < HOST_SIDE_STREAMS; ++k) {
        HIP_TRY(hipStreamWaitEvent(g->side[k], g->host_begin, 0));
        compute[1 + k] = g->side[k];
    }
    constexpr size_t NCS = 1 + HOST_SIDE_STREAMS;
    hipStream_t converted_on[HOST_MAX_PIECES] = {};     // the stream a piece's columns were converted on (job_ingest)
    size_t last_key_piece = np;                         // the piece whose arrival completes the key columns
    for (size_t i = 0; i < np; ++i)
        if (b.pieces[i].cols & COLS_KEYS) last_key_piece = i;
    struct deferred { size_t first, count; hipStream_t cs; };
    // ranges whose hashes cannot be queued yet: a wire call hashes behind its key kernels; and a call that was expected to
    // hash with the head launch but does not after all (job_begin could not set the key tables up) reads u, which then
    // travels last
    std::vector<deferred> waiting;
    const bool needs_late = late && !J.split;
    size_t last_late_piece = np;
    for (size_t i = 0; i < np; ++i)
        if (b.pieces[i].cols & COLS_LATE) last_late_piece = i;
    auto hashes_blocked = [&](size_t i) {
        return (job_hash_needs_keys(J) && !J.keys_queued) || (needs_late && i < last_late_piece);
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
        hipStream_t cs = compute[i % NCS];
        if (i % NCS) J.side[i % NCS - 1] = cs;
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
            for (size_t j = 0; j <= i; ++j)
                if (b.pieces[j].cols & COLS_KEYS) HIP_TRY(hipStreamWaitEvent(sl->key_stream, g->chunk_done[j], 0));
            if (int rc = job_keys(J)) return rc;
        }
        if (!waiting.empty() && !hashes_blocked(i)) {
            for (const deferred& d : waiting) {
                if (needs_late) HIP_TRY(hipStreamWaitEvent(d.cs, g->chunk_up[last_late_piece], 0));
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

int run_host(const host_col* cols, size_t n_cols, size_t n, uint8_t* status, uint64_t tally[4], call_builder build, bool keys_gate_hashes) {
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
        plan_pieces(b.plans[0], b.largest_bytes, b.hi - b.lo, row_keys, row_rest, row_late, keys_gate_hashes, false);
        plan_pieces(b.plans[1], b.largest_bytes, b.hi - b.lo, row_keys, row_rest, row_late, keys_gate_hashes, true);
        b.staging_threads = staging_threads;
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

}  // namespace

extern "C" {

int jjs_abi_version(void) { return 4; }
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
    std::lock_guard<std::mutex> lock(L.mu);
    shutdown_locked();
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
    if (sl->wire) {
        HIP_TRY(hipDeviceSynchronize());        // earlier launches may still read the old buffer
        HIP_TRY(hipFree(sl->wire));
        sl->wire = nullptr; sl->wire_items = 0;
    }
    size_t cap = n < 4096 ? 4096 : n;
    HIP_TRY(hipMalloc(&sl->wire, cap * (4 * 64 + 16 + 2 * 48)));
    sl->wire_items = cap;
    return JJS_OK;
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
// a host-buffer call: blocking
static int host_call(call_builder build, const host_col* cols, size_t n_cols, size_t n, uint8_t* status, uint64_t tally[4],
                     bool keys_gate_hashes = false) {
    std::lock_guard<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    return no_throw([&] { return run_host(cols, n_cols, n, status, tally, build, keys_gate_hashes); });
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
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 64, COLS_REST}, {PK, 64, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_affine_single, cols, 4, n, status, tally);
}
int jjs_verify_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                      const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 64, COLS_REST}, {Rp, 64, COLS_REST}, {PK, 64, COLS_KEYS}, {PKp, 64, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_affine_double, cols, 6, n, status, tally);
}
int jjs_verify_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                      size_t n, uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 64, COLS_REST}, {PK, 64, COLS_KEYS}, {Gen, 64, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_affine_vargen, cols, 5, n, status, tally);
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
    const host_col cols[] = {{sig, 64, COLS_REST}, {pk, 32, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_wire_single, cols, 3, n, status, tally, true);
}
int jjs_verify_double_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{sig, 96, COLS_REST}, {pk, 64, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_wire_double, cols, 3, n, status, tally, true);
}
int jjs_verify_vargen_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{sig, 64, COLS_REST}, {pk, 64, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_wire_vargen, cols, 3, n, status, tally, true);
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
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 96, COLS_REST}, {PK, 96, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_ext_single, cols, 4, n, status, tally);
}
int jjs_verify_double_ext(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK, const uint8_t* PKp,
                          const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 96, COLS_REST}, {Rp, 96, COLS_REST}, {PK, 96, COLS_KEYS}, {PKp, 96, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_ext_double, cols, 6, n, status, tally);
}
int jjs_verify_vargen_ext(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m, size_t n,
                          uint8_t* status, uint64_t tally[4]) {
    const host_col cols[] = {{u, 32, COLS_LATE}, {R, 96, COLS_REST}, {PK, 96, COLS_KEYS}, {Gen, 96, COLS_KEYS}, {m, 32, COLS_REST}};
    return host_call(build_ext_vargen, cols, 5, n, status, tally);
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
        if (c.key_stream) note_key_feedback();
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
    // SAFE tags of the transcript are computed here (csrc/safe_tag.h) and travel with the call.
    bool any_long = false;
    for (size_t t = 0; t < n_transcripts; ++t) {
        if (offsets_host[t + 1] < offsets_host[t]) return fail(JJS_ERR_ARG, "transcript %zu: offsets must not decrease", t);
        const uint64_t cnt = offsets_host[t + 1] - offsets_host[t];
        if (cnt > JJS_MSIG_PARTICIPANTS_LIMIT)
            return fail(JJS_ERR_ARG, "transcript %zu: %llu participants (the hash transcripts are indexed with 32 bits: at most %u)", t,
                        (unsigned long long)cnt, (unsigned)JJS_MSIG_PARTICIPANTS_LIMIT);
        any_long = any_long || cnt > JJS_MSIG_MAX_PARTICIPANTS;
    }
    const size_t n = offsets_host[n_transcripts];
    if ((n && !all_ok(z, PK, R, S)) || !all_ok(m, agg_pk, sig_u, sig_R) || (n && !share_status)) return fail(JJS_ERR_ARG, "null or misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n > g->msig_items || n_transcripts > g->msig_transcripts) {
        if (g->msig) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(g->msig)); g->msig = nullptr; }
        size_t ci = n < 4096 ? 4096 : n, ct = n_transcripts < 1024 ? 1024 : n_transcripts;
        HIP_TRY(hipMalloc(&g->msig, ci * 4 * (1 + 8 + 2 * EXT_WORDS) + ct * 4 * (16 + 1 + 18) + 64));
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
    P.long_tags = any_long ? d_long : nullptr;
    big_slot();
    if (int rc = begin_shared(s)) return rc;
    std::vector<uint32_t> long_tags;            // outlives the (pageable, hence staged) copy below
    auto queue = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_off, offsets_host, (n_transcripts + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (any_long) {
            long_tags.assign(n_transcripts * 18, 0u);
            for (size_t t = 0; t < n_transcripts; ++t) {
                const uint32_t cnt = offsets_host[t + 1] - offsets_host[t];
                if (cnt <= JJS_MSIG_MAX_PARTICIPANTS) continue;
                safe_tag_limbs(2u + 2u * cnt, JJS_Q_WORDS, &long_tags[18 * t]);
                safe_tag_limbs(3u + 4u * cnt, JJS_Q_WORDS, &long_tags[18 * t + 9]);
            }
            HIP_TRY(hipMemcpyAsync(d_long, long_tags.data(), long_tags.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));       // rare path (a transcript of more than 256 participants): keep it simple
        }
        for (int pass = 0; pass < 7; ++pass) {
            const size_t count = (pass == 0 || pass == 2 || pass == 4 || pass == 6) ? n_transcripts : n;
            hipLaunchKernelGGL(msig_kernel, dim3(grid_for(g->grid_msig, count)), dim3(BLOCK), 0, s, P, pass);
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
    g_force_positions = ((which >> 4) & 15) == 4 || ((which >> 4) & 15) == 8 ? ((which >> 4) & 15) : 0;    // 0x42 / 0x82: latency path, 4 / 8 pieces
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
