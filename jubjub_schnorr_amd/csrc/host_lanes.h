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
    HIP_TRY(hipMemcpyAsync(lane.dev, lane.pinned, Y.in_bytes, hipMemcpyHostToDevice, lane.stream));      // one copy: the area is at most ~2 x what is in it
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
// Waiting for another thread's word: poll the epoch for LANE_SPIN_US (the waits are fractions of a millisecond, and a sleeping
// thread would come back too late to share the next launch), then sleep on the condition variable.
static void lane_wait_change(std::unique_lock<std::mutex>& lock, device_state* dev) {
    const uint64_t e = dev->lane_epoch.load(std::memory_order_relaxed);
    lock.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned turn = 0; dev->lane_epoch.load(std::memory_order_acquire) == e; ++turn) {
        if ((turn & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(JJS_LANE_SPIN_US)) break;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        _mm_pause();
#endif
    }
    lock.lock();
    L.lane_cv.wait(lock, [&] { return dev->lane_epoch.load(std::memory_order_relaxed) != e; });
}
static void lane_changed(device_state* dev) {          // under the engine's mutex
    dev->lane_epoch.fetch_add(1, std::memory_order_release);
    L.lane_cv.notify_all();
}
// One call.  It joins the lane that is filling for its shape or opens one; copies its columns into the lane's pinned area
// (outside the mutex); then whichever member finds the lane complete (nobody copying) and no other launch of the shape
// running sends it off and waits for it; the others wait for that member's word.
static int lane_call(int scheme, int format, const uint8_t* const* cols, size_t n, uint8_t* status, uint64_t tally[4]) {
    const call_shape& S = SHAPES[scheme][format];
    const bool combinable = n <= COMBINE_MAX_CALL_ITEMS;
    device_state* dev = nullptr;
    host_lane* lane = nullptr;
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
            lane->state = host_lane::OPEN; lane->scheme = scheme; lane->format = format; lane->combinable = combinable;
            lane->cap = cap; lane->items = 0; lane->copying = 0; lane->members = 0; lane->rc = JJS_OK; lane->err[0] = 0;
            lane->gather_until = std::chrono::steady_clock::now();
        }
        if (lane) break;
        lane_wait_change(lock, dev);
        if (L.devs.empty() || check_ready() != JJS_OK || g != dev) return fail(JJS_ERR_NOT_INIT, "the engine was shut down during the call");
    }
    const size_t first = lane->items;
    lane->items += n;
    ++lane->members;
    ++lane->copying;
    if (lane->items == lane->cap) lane->combinable = false;        // full
    const lane_layout Y = lane_layout_for(S, lane->cap);
    lock.unlock();
    for (size_t k = 0; k < S.n_cols; ++k) memcpy(lane->pinned + Y.off[k] + first * S.col[k].width, cols[k], n * S.col[k].width);
    lock.lock();
    if (--lane->copying == 0) lane_changed(dev);
    while (lane->state != host_lane::DONE) {
        // combined launches of one shape run one at a time (everything else: side by side)
        bool shape_busy = false;
        if (combinable)
            for (const host_lane& l : dev->lanes)
                shape_busy = shape_busy || (&l != lane && l.state == host_lane::LAUNCHED && l.combinable_shape && l.scheme == scheme && l.format == format);
        if (lane->state == host_lane::OPEN && lane->copying == 0 && !shape_busy) {
            const auto until = lane->gather_until;
            if (lane->combinable && std::chrono::steady_clock::now() < until) {
                // callers that are on their way may still join (a short spin: the timers of a sleeping wait are coarser than this)
                lock.unlock();
                while (std::chrono::steady_clock::now() < until) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
                    _mm_pause();
#endif
                }
                lock.lock();
                continue;
            }
            lane->state = host_lane::LAUNCHED;            // closed: its items are final
            lane->combinable_shape = combinable;
            ++dev->stats[JJS_PATH_LANE_LAUNCHES];
            dev->stats[JJS_PATH_LANE_CALLS] += lane->members;
            lane_changed(dev);
            lock.unlock();
            int rc = no_throw([&] { return lane_launch(dev, *lane, S); });
            // whatever was queued drains before anybody touches the lane again, also after a failure
            const hipError_t e = rc == JJS_OK ? lane_wait(*lane) : hipStreamSynchronize(lane->stream);
            if (rc == JJS_OK && e != hipSuccess) rc = fail(JJS_ERR_HIP, "waiting for the launch: %s", hipGetErrorString(e));
            lock.lock();
            lane->rc = rc;
            if (rc != JJS_OK) snprintf(lane->err, sizeof(lane->err), "%s", t_err);
            lane->state = host_lane::DONE;
            if (combinable) {
                dev->lane_last_items[scheme][format] = lane->items;
                // the lane that filled behind this launch waits a moment for the callers this launch is about to release
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(COMBINE_WINDOW_US);
                for (host_lane& l : dev->lanes)
                    if (l.state == host_lane::OPEN && l.scheme == scheme && l.format == format) l.gather_until = until;
            }
            lane_changed(dev);
            break;
        }
        lane_wait_change(lock, dev);
    }
    int rc = lane->rc;
    if (rc != JJS_OK) rc = fail(rc, "%s", lane->err);
    else {
        const uint8_t* st = lane->pinned + Y.status_off + first;
        lock.unlock();
        if (status) memcpy(status, st, n);
        if (tally) {
            uint64_t t[256] = {};
            for (size_t i = 0; i < n; ++i) ++t[st[i]];
            for (int k = 0; k < 4; ++k) tally[k] = t[k];
        }
        lock.lock();
    }
    if (--lane->members == 0) { lane->state = host_lane::FREE; lane_changed(dev); }
    return rc;
}

