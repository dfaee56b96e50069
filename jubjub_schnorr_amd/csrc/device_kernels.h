// Part of jjs_gpu.hip (included inside its anonymous namespace): every __global__ entry of the library.  The arithmetic
// lives in the headers they call (verify_core.h, key_tables.h, small_batch.h, decode.h, normalize.h, sign_core.h,
// multisig_core.h); a kernel here is a grid-stride loop, a role by block index, and the wave-level reductions.
#pragma once
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
// What a call finds cleared on its own stream (counters of the caller, the queue of the resolve pass, the flags of a wire or
// ext call), in ONE launch instead of one fill each: every small dependent launch at the head of a call is 15-50 us.
struct clear_params {
    void* p[3];
    uint64_t bytes[3];
};
__global__ __launch_bounds__(BLOCK) void clear_kernel(clear_params C) {
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x, total = (uint64_t)gridDim.x * BLOCK;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint8_t* const p = static_cast<uint8_t*>(C.p[k]);
        const uint64_t bytes = C.bytes[k];
        if (!p || !bytes) continue;
        const uint64_t head = (uint64_t)(-(intptr_t)p & 15) < bytes ? (uint64_t)(-(intptr_t)p & 15) : bytes;     // up to 16-byte alignment
        const uint64_t body = (bytes - head) / 16;
        uint4* const q = reinterpret_cast<uint4*>(p + head);
        for (uint64_t i = t; i < body; i += total) q[i] = uint4{0u, 0u, 0u, 0u};
        const uint64_t tail0 = head + body * 16;
        for (uint64_t i = t; i < head; i += total) p[i] = 0;
        for (uint64_t i = tail0 + t; i < bytes; i += total) p[i] = 0;
    }
}
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
// The chain of bases and `is_valid` of every key.  One lane per key does both (kt_chain_key); where the batch waits for the
// chains (K.quad_chains, set by job_keys) four lanes per key share the doublings (kt_chain_key_quad) and a fifth runs
// `is_valid`: the grid covers 5 x n_cols x max_keys lanes then.
__global__ __launch_bounds__(BLOCK, 2) void key_chain_kernel(key_params K) {
    const int w = (int)K.counters[2];
    if (!w) return;
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x, keys = K.n_cols * K.max_keys;
    // a key found valid joins the list of the keys that get tables (key_table_kernel)
    if (!K.quad_chains) {
        const uint32_t c = t / K.max_keys, id = t % K.max_keys;
        if (c < K.n_cols && id < K.counters[c]) {
            const key_column C = kt_col(K, (int32_t)c);
            if (kt_chain_key(C, id, w)) C.valid_ids[atomicAdd(&K.counters[5 + c], 1u)] = id;
        }
    } else if (t < 4 * keys) {
        const uint32_t q = t >> 2, c = q / K.max_keys, id = q % K.max_keys;          // the same for the four lanes of a quad
        if (id < K.counters[c]) kt_chain_key_quad(kt_col(K, (int32_t)c), id, w, t & 3u);
    } else if (t < 5 * keys) {
        const uint32_t q = t - 4 * keys, c = q / K.max_keys, id = q % K.max_keys;
        if (id < K.counters[c]) {
            const key_column C = kt_col(K, (int32_t)c);
            if (kt_key_flags(C, id)) C.valid_ids[atomicAdd(&K.counters[5 + c], 1u)] = id;
        }
    }
}
// the grid covers max_keys x KT_MAX_POSITIONS lanes per column; a batch with wide windows has fewer of both, and only the
// keys whose point is valid get tables (kt_finish_item)
__global__ __launch_bounds__(BLOCK, 2) void key_table_kernel(key_params K) {
    const int w = (int)K.counters[2];
    if (!w) return;
    const uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t positions = (uint32_t)kt_positions(w);
    const uint64_t per_col = (uint64_t)K.max_keys * positions;
    const uint32_t c = (uint32_t)(t / per_col), j = (uint32_t)((t % per_col) / positions), pos = (uint32_t)(t % positions);
    if (c < K.n_cols && j < K.counters[5 + c]) {            // the j-th VALID key of the column
        const key_column C = kt_col(K, (int32_t)c);
        kt_table_lane(C, C.valid_ids[j], pos, w);
    }
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

// Second pass: the queued items, densely packed over the lanes, P.resolve_lanes (resolve_lanes_keyed) adjacent lanes per item
// (one point each; see verify_item / resolve_point).
__global__ __launch_bounds__(BLOCK) void resolve_kernel(verify_params P) {
    const uint64_t gtid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    const uint64_t count = *P.pending_count;
    const uint32_t L = keyed_mode(P) ? P.resolve_lanes_keyed : P.resolve_lanes;      // a lane per point that still needs its test
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
#if defined(JJS_AB_SMALL_ONLY_HASH)        // timing experiments: the hash lanes alone (results are garbage)
    return;
#endif
    if (cb < S.positions * chain_blocks_per_pos) {
        const uint32_t k = S.positions - 1 - cb / chain_blocks_per_pos;            // block-uniform position
        const uint64_t r = (uint64_t)(cb % chain_blocks_per_pos) * BLOCK + threadIdx.x;
        const uint32_t per_item = 2 * S.V.n_eq;                                     // (equation, PK | R)
        if (S.quad_chains) {                                                        // four adjacent lanes per chain
            const uint64_t q = r >> 2;
            if (q < n * per_item) sb_chain_lane_quad(S, q / per_item, (uint32_t)(q % per_item) >> 1, (uint32_t)q & 1u, k, (uint32_t)r & 3u);
            return;
        }
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
    if (pos >= 8) acc = sb_add(acc, shfl_xor_ext(acc, 4));
    if (pos == 16) acc = sb_add(acc, shfl_xor_ext(acc, 8));
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
    uint32_t n_src;
    uint32_t split;       // 1 (decode_points_kernel): one lane per (item, source) instead of one per item -- the calls of the latency
                          // path, which wait for the one square root of a point (~330 dependent products), not for how many there are
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
// The same, one lane per (item, source): the calls of the latency path (decode_params::split).  A kernel of its own: as a branch
// of decode_kernel it cost the resident 2^20 wire batch 4.8 % (profiles/r04_wire_small_ab.jsonl).
__global__ __launch_bounds__(BLOCK) void decode_points_kernel(decode_params P) {
    const uint64_t total = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < P.n * P.n_src; j += total) {
        const uint64_t item = P.first + j / P.n_src;
        const uint32_t k = (uint32_t)(j % P.n_src);
        // source and destination chosen field by field: indexing the kernel-argument struct with a run-time index would
        // make the compiler copy it to scratch memory
        fe_src src = P.src[0];
        uint8_t* out = P.out[0];
#pragma unroll
        for (uint32_t c = 1; c < 4; ++c) {
            const bool me = k == c;
            src.base = me ? P.src[c].base : src.base; src.stride = me ? P.src[c].stride : src.stride; src.off = me ? P.src[c].off : src.off;
            out = me ? P.out[c] : out;
        }
        const decoded_point d = decompress_point(load_words(src, item), P.dlog);
        store_words(out, 2 * item, d.u);
        store_words(out, 2 * item + 1, d.v);
        if (P.bad && !d.ok) P.bad[item] = 1;          // several lanes of an item may write the same 1
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
    // the passes with a hash chain, on eight lanes per item when the call has few of them (whole groups of eight adjacent lanes
    // enter and leave the loop together: the shuffles of a group stay inside it)
    const uint32_t hl = (pass == 1 || pass == 2 || pass == 4) ? P.hash_lanes : 1u;
    const int coop = hl > 1 ? (int)(gtid % hl) : -1;
    for (uint64_t i = gtid / hl; i < count; i += total / hl) {
        switch (pass) {
        case 0: msig_map_item(P, (uint32_t)i); break;
        case 1: msig_delin_item(P, i, ws, coop); break;
        case 2: msig_agg_item(P, (uint32_t)i, coop); break;
        case 3: msig_commit_item(P, i, ws); break;
        case 4: msig_final_item(P, (uint32_t)i, coop); break;
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
