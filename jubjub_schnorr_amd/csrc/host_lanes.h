// Part of jjs_gpu.hip (included behind the table of call shapes): the staging lanes of the small blocking host-buffer calls --
// a call of at most LANE_MAX_ITEMS items joins or opens a lane, copies its arrays into the lane's pinned area outside the
// engine's mutex, and the lane goes to the device as one launch for all its members (engine_state.h host_lane has the why).
#pragma once
// ---- host lanes: host-buffer calls of at most LANE_MAX_ITEMS items (see host_lane) ------------------------------------
// layout of a lane's two areas (the same in pinned host memory and on the device): one array per column, sized for `cap`
// items, then the statuses
struct lane_layout {
    size_t off[8];
    size_t in_bytes, status_off, total;
};
static lane_layout lane_layout_for(const call_shape& S, size_t cap) {
    lane_layout Y{};
    size_t p = 0;
    for (size_t k = 0; k < S.n_cols; ++k) { Y.off[k] = p; p += pad256(cap * S.col[k].width); }
    Y.in_bytes = p;
    Y.status_off = p; p += pad256(cap);
    Y.total = p;
    return Y;
}
// Items a new lane is laid out for: a call too large to combine gets exactly its own; else twice what the last combined launch
// of the shape carried (the area is uploaded whole, so it should not be much larger than what will be in it).
static size_t lane_cap_for(const device_state* dev, int scheme, int format, size_t n) {
    if (n > COMBINE_MAX_CALL_ITEMS) return n;
    size_t want = 2 * (dev ? dev->lane_last_items[scheme][format] : COMBINE_CAP_ITEMS);
    if (want < n) want = n;
    size_t cap = 256;
    while (cap < want) cap <<= 1;
    return cap < COMBINE_CAP_ITEMS ? cap : COMBINE_CAP_ITEMS;
}
static int ensure_lane(host_lane& lane, size_t bytes) {
    if (!lane.stream) HIP_TRY(hipStreamCreateWithFlags(&lane.stream, hipStreamNonBlocking));
    if (!lane.done) HIP_TRY(hipEventCreateWithFlags(&lane.done, hipEventDisableTiming));
    if (bytes > lane.dev_bytes) {
        const size_t cap = grown(bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes);
        if (int rc = regrow(lane.dev, lane.dev_bytes, lane.dev_bytes, cap, cap)) return rc;
    }
    if (bytes > lane.pinned_bytes) {
        const size_t cap = grown(bytes < (size_t(1) << 20) ? (size_t(1) << 20) : bytes);
        uint8_t* fresh = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&fresh), cap, hipHostMallocDefault));
        retire(lane.pinned, true, lane.pinned_bytes);
        lane.pinned = fresh;
        lane.pinned_bytes = cap;
    }
    return JJS_OK;
}
// The end of a lane launch: small calls last half a millisecond, and a thread that sleeps on the stream pays the wake-up on
// top (tens to hundreds of microseconds on an idle core), so it polls the launch's event for LANE_SPIN_US first.
#ifndef JJS_LANE_SPIN_US
#define JJS_LANE_SPIN_US 2000
#endif
static hipError_t lane_wait(host_lane& lane) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned turn = 0;; ++turn) {
        const hipError_t e = hipEventQuery(lane.done);
        if (e != hipErrorNotReady) return e;
        if ((turn & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(JJS_LANE_SPIN_US)) break;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        _mm_pause();
#endif
    }
    (void)hipGetLastError();
    return hipStreamSynchronize(lane.stream);
}
// Upload, kernels and download of the lane's launch (its `items` items).  Called WITHOUT the engine's mutex (the lane is in
// state LAUNCHED: nobody else touches it); takes the mutex for the part that uses the engine's state (slot, launches).
static int lane_launch(device_state* dev, host_lane& lane, const call_shape& S) {
    const lane_layout Y = lane_layout_for(S, lane.cap);
    const size_t n = lane.items;
    HIP_TRY(hipSetDevice(dev->device));
    // one copy of the whole area (at most ~2 x what is in it): a copy per column of the part in use costs more than the bytes it
    // saves (+ 15-25 us per launch of 1 024-4 096 signatures, profiles/r04_lane_trace.jsonl)
    HIP_TRY(hipMemcpyAsync(lane.dev, lane.pinned, Y.in_bytes, hipMemcpyHostToDevice, lane.stream));
    {
        std::lock_guard<std::mutex> lock(L.mu);
        g = dev;
        const void* in[8];
        for (size_t k = 0; k < S.n_cols; ++k) in[k] = lane.dev + Y.off[k];
        staged_call C;
        // no device tally: every member counts its own statuses
        if (int r = S.build(in, n, lane.dev + Y.status_off, nullptr, lane.stream, C)) return r;
        if (int r = launch_staged(C, lane.stream)) return r;
    }
    HIP_TRY(hipMemcpyAsync(lane.pinned + Y.status_off, lane.dev + Y.status_off, n, hipMemcpyDeviceToHost, lane.stream));
    HIP_TRY(hipEventRecord(lane.done, lane.stream));
    return JJS_OK;
}
#if defined(JJS_LANE_TRACE)
// investigation builds only: where a lane launch spends its time, summed over the launches and printed by jjs_shutdown
struct lane_trace_sums { uint64_t n, members, items, open_ready_ns, ready_queued_ns, queued_done_ns, done_free_ns; };
static lane_trace_sums g_lane_trace = {};
static inline uint64_t trace_ns(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
}
static void lane_trace_dump() {
    const lane_trace_sums& t = g_lane_trace;
    if (!t.n) return;
    fprintf(stderr, "{\"lane_trace\": {\"launches\": %llu, \"members\": %.2f, \"items\": %.0f, \"open_to_ready_us\": %.1f, \"ready_to_queued_us\": %.1f, "
            "\"queued_to_done_us\": %.1f, \"done_to_free_us\": %.1f}}\n", (unsigned long long)t.n, (double)t.members / t.n, (double)t.items / t.n,
            1e-3 * t.open_ready_ns / t.n, 1e-3 * t.ready_queued_ns / t.n, 1e-3 * t.queued_done_ns / t.n, 1e-3 * t.done_free_ns / t.n);
    g_lane_trace = lane_trace_sums{};
}
#endif
// A polling ticket: at most LANE_MAX_SPINNERS threads poll for a lane at a time (engine_state.h).
struct spin_ticket {
    device_state* dev;
    bool held;
    explicit spin_ticket(device_state* d) : dev(d), held(d->lane_spinners.fetch_add(1, std::memory_order_relaxed) < LANE_MAX_SPINNERS) {
        if (!held) dev->lane_spinners.fetch_sub(1, std::memory_order_relaxed);
    }
    ~spin_ticket() { if (held) dev->lane_spinners.fetch_sub(1, std::memory_order_relaxed); }
    spin_ticket(const spin_ticket&) = delete;
    spin_ticket& operator=(const spin_ticket&) = delete;
};
static inline void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    _mm_pause();
#endif
}
// Waiting for a change of any lane (a lane leader waiting for its turn; a caller for which no lane is free): poll the epoch
// for LANE_SPIN_US with a ticket (the waits are fractions of a millisecond, and a sleeping thread would come back too late
// to share the next launch), then -- or at once, without a ticket -- sleep on the condition variable.
static void lane_wait_change(std::unique_lock<std::mutex>& lock, device_state* dev, bool leader) {
    const uint64_t e = dev->lane_epoch.load(std::memory_order_relaxed);
    lock.unlock();
    {
        spin_ticket ticket(dev);
        if (ticket.held || leader) {                    // a leader always polls: its members wait for it (at most one per lane)
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned turn = 0; dev->lane_epoch.load(std::memory_order_acquire) == e; ++turn) {
                if ((turn & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(JJS_LANE_SPIN_US)) break;
                cpu_relax();
            }
        }
    }
    lock.lock();
    L.lane_cv.wait(lock, [&] { return dev->lane_epoch.load(std::memory_order_relaxed) != e; });
}
static void lane_changed(device_state* dev) {          // under the engine's mutex
    dev->lane_epoch.fetch_add(1, std::memory_order_release);
    L.lane_cv.notify_all();
}
// A member that is not the leader waits for the end of the lane's launch, without the engine's mutex: `gen` is what done_gen
// was when it joined.  Polls with a ticket, else sleeps on the word itself.
static void lane_await_done(device_state* dev, host_lane& lane, uint32_t gen) {
    {
        spin_ticket ticket(dev);
        if (ticket.held) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned turn = 0; lane.done_gen.load(std::memory_order_acquire) == gen; ++turn) {
                if ((turn & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(JJS_LANE_SPIN_US)) break;
                cpu_relax();
            }
        }
    }
    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the futex word is the atomic itself");
    while (lane.done_gen.load(std::memory_order_acquire) == gen)
        (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&lane.done_gen), FUTEX_WAIT_PRIVATE, gen, nullptr, nullptr, 0);
}
static void lane_publish_done(host_lane& lane) {        // under the engine's mutex, state already DONE
    lane.done_gen.fetch_add(1, std::memory_order_release);
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(&lane.done_gen), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
// One call.  It joins the lane that is filling for its shape or opens one -- and is then that lane's LEADER; copies its columns
// into the lane's pinned area (outside the mutex).  The leader waits until the lane is complete (nobody copying) and no other
// combined launch of the shape is running, sends the lane off, waits for the device and publishes the outcome; the other
// members wait for that word only (lane_await_done).
static int lane_call(int scheme, int format, const uint8_t* const* cols, size_t n, uint8_t* status, uint64_t tally[4]) {
    const call_shape& S = SHAPES[scheme][format];
    const bool combinable = n <= COMBINE_MAX_CALL_ITEMS;
    device_state* dev = nullptr;
    host_lane* lane = nullptr;
    bool leader = false;
    std::unique_lock<std::mutex> lock(L.mu);
    if (int rc = check_ready()) return rc;
    dev = g;
    for (;;) {
        host_lane* free_lane = nullptr;
        for (host_lane& l : dev->lanes) {
            if (combinable && l.state == host_lane::OPEN && l.combinable && l.scheme == scheme && l.format == format && l.items + n <= l.cap) { lane = &l; break; }
            if (l.state == host_lane::FREE && !free_lane) free_lane = &l;
        }
        if (!lane && free_lane) {
            const size_t cap = lane_cap_for(dev, scheme, format, n);
            if (int rc = ensure_lane(*free_lane, lane_layout_for(S, cap).total)) return rc;
            lane = free_lane;
            leader = true;
            lane->state = host_lane::OPEN; lane->scheme = scheme; lane->format = format; lane->combinable = combinable;
            lane->cap = cap; lane->items = 0; lane->copying = 0; lane->members = 0; lane->rc = JJS_OK; lane->err[0] = 0;
            lane->gather_until = std::chrono::steady_clock::now();
            lane->expect = 0;
#if defined(JJS_LANE_TRACE)
            lane->trace_open = lane->gather_until;
#endif
            if (JJS_COMBINE_EXPECT && combinable && dev->lane_last_members[scheme][format] > 1) {      // behind a launch of several calls: see COMBINE_WINDOW_US
                const auto until = dev->lane_last_done[scheme][format] + std::chrono::microseconds(COMBINE_WINDOW_US);
                if (lane->gather_until < until) { lane->gather_until = until; lane->expect = dev->lane_last_members[scheme][format]; }
            }
        }
        if (lane) break;
        lane_wait_change(lock, dev, false);
        if (L.devs.empty() || check_ready() != JJS_OK || g != dev) return fail(JJS_ERR_NOT_INIT, "the engine was shut down during the call");
    }
    const size_t first = lane->items;
    const uint32_t gen = lane->done_gen.load(std::memory_order_relaxed);
    lane->items += n;
    ++lane->members;
    ++lane->copying;
    if (lane->items == lane->cap) lane->combinable = false;        // full
    const lane_layout Y = lane_layout_for(S, lane->cap);
    lock.unlock();
    for (size_t k = 0; k < S.n_cols; ++k) memcpy(lane->pinned + Y.off[k] + first * S.col[k].width, cols[k], n * S.col[k].width);
    lock.lock();
    if (--lane->copying == 0 && !leader) lane_changed(dev);         // the leader may be waiting for this
    if (!leader) {
        lock.unlock();
        lane_await_done(dev, *lane, gen);
    }
    while (leader) {
        // combined launches of one shape run one at a time (everything else: side by side) -- unless the lane holds all the
        // latency path takes: nobody can join it any more, and two such launches fill the device better than one
        bool shape_busy = false;
        if (combinable && !(lane->cap == COMBINE_CAP_ITEMS && lane->items > COMBINE_CAP_ITEMS - COMBINE_CAP_ITEMS / 8))
            for (const host_lane& l : dev->lanes)
                shape_busy = shape_busy || (&l != lane && l.state == host_lane::LAUNCHED && l.combinable_shape && l.scheme == scheme && l.format == format);
        if (lane->copying == 0 && !shape_busy) {
            const auto until = lane->gather_until;
            if (lane->combinable && lane->members < lane->expect && std::chrono::steady_clock::now() < until) {
                // callers that are on their way may still join (a short spin: the timers of a sleeping wait are coarser than this)
                const unsigned expect = lane->expect;
                lock.unlock();
                while (lane->members.load(std::memory_order_relaxed) < expect && std::chrono::steady_clock::now() < until) cpu_relax();
                lock.lock();
                if (lane->members < expect) lane->expect = 0;          // the window has passed: whoever has joined, joined
                continue;
            }
            lane->state = host_lane::LAUNCHED;            // closed: its items are final
            lane->combinable_shape = combinable;
            ++dev->stats[JJS_PATH_LANE_LAUNCHES];
            dev->stats[JJS_PATH_LANE_CALLS] += lane->members;
            lane_changed(dev);
            lock.unlock();
#if defined(JJS_LANE_TRACE)
            const auto t_ready = std::chrono::steady_clock::now();
#endif
            int rc = no_throw([&] { return lane_launch(dev, *lane, S); });
#if defined(JJS_LANE_TRACE)
            const auto t_queued = std::chrono::steady_clock::now();
#endif
            // whatever was queued drains before anybody touches the lane again, also after a failure
            const hipError_t e = rc == JJS_OK ? lane_wait(*lane) : hipStreamSynchronize(lane->stream);
            if (rc == JJS_OK && e != hipSuccess) rc = fail(JJS_ERR_HIP, "waiting for the launch: %s", hipGetErrorString(e));
            lock.lock();
#if defined(JJS_LANE_TRACE)
            lane->trace_done = std::chrono::steady_clock::now();
            ++g_lane_trace.n; g_lane_trace.members += lane->members; g_lane_trace.items += lane->items;
            g_lane_trace.open_ready_ns += trace_ns(lane->trace_open, t_ready); g_lane_trace.ready_queued_ns += trace_ns(t_ready, t_queued);
            g_lane_trace.queued_done_ns += trace_ns(t_queued, lane->trace_done);
#endif
            lane->rc = rc;
            if (rc != JJS_OK) snprintf(lane->err, sizeof(lane->err), "%s", t_err);
            lane->state = host_lane::DONE;
            if (combinable) {
                dev->lane_last_items[scheme][format] = lane->items;
                dev->lane_last_members[scheme][format] = lane->members;
                const auto now = std::chrono::steady_clock::now();
                dev->lane_last_done[scheme][format] = now;
                // the lane that filled behind this launch waits a moment for the callers this launch is about to release
                for (host_lane& l : dev->lanes)
                    if (l.state == host_lane::OPEN && l.scheme == scheme && l.format == format) {
                        l.gather_until = now + std::chrono::microseconds(COMBINE_WINDOW_US);
                        l.expect = JJS_COMBINE_EXPECT ? l.members + lane->members : UINT_MAX;
                    }
            }
            lane_publish_done(*lane);
            lane_changed(dev);
            lock.unlock();
            break;
        }
        lane_wait_change(lock, dev, true);
    }
    // the launch has ended (the outcome was written before done_gen moved) and the lane stays as it is until its last member
    // has left: the statuses are read without the mutex
    int rc = lane->rc;
    if (rc != JJS_OK) rc = fail(rc, "%s", lane->err);
    else {
        const uint8_t* st = lane->pinned + Y.status_off + first;
        if (status) memcpy(status, st, n);
        if (tally) {
            uint64_t t[256] = {};
            for (size_t i = 0; i < n; ++i) ++t[st[i]];
            for (int k = 0; k < 4; ++k) tally[k] = t[k];
        }
    }
    lock.lock();
    if (--lane->members == 0) {
#if defined(JJS_LANE_TRACE)
        g_lane_trace.done_free_ns += trace_ns(lane->trace_done, std::chrono::steady_clock::now());
#endif
        lane->state = host_lane::FREE;
        lane_changed(dev);
    }
    return rc;
}

