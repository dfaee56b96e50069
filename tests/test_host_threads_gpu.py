"""The blocking host-buffer entry points under the reference's own call pattern: few signatures per call, many host
threads (the reference API verifies one item per call, /root/reference/src/keys/public.rs:114-118; its callers are
services with a thread per request).  Calls of at most 131 072 items take a staging lane each and hold the engine's
mutex only while they are queued (csrc/jjs_gpu.hip lane_call), so they run side by side on the device.

Also here: what keeps an asynchronous call asynchronous (no device-wide wait when a buffer grows, jjs_reserve), and the
ordering of the key-table decision word between two host-buffer wire calls in one slot."""
import threading
import time

import numpy as np
import pytest

from helpers import ARG_ORDER, IDENT, batch_to_extended, make_batch, oracle_verify, to_wire

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    import jubjub_schnorr_amd as jjs
    return jjs.engine()


def _call(eng, scheme, fmt, args):
    if fmt == "affine":
        return eng.verify(scheme, *args)
    return eng.verify_ext(scheme, *args) if fmt == "ext" else eng.verify_wire(scheme, *args)


def _mixed_work(n):
    specs = [("single", "affine"), ("double", "affine"), ("vargen", "wire"), ("single", "ext"), ("single", "wire"), ("double", "ext")]
    work = []
    for i, (scheme, fmt) in enumerate(specs):
        b = make_batch(scheme, n, seed=4100 + i, n_keys=16)
        want = oracle_verify(scheme, b)
        args = {"affine": [b[k] for k in ARG_ORDER[scheme]], "ext": batch_to_extended(scheme, b, seed=i),
                "wire": list(to_wire(scheme, b))}[fmt]
        work.append((scheme, fmt, [np.ascontiguousarray(a) for a in args], want))
        assert set(want.tolist()) >= {0, 1, 2}
    return work


def test_four_threads_of_small_host_calls_against_the_oracle(eng, tmp_path):
    """4 host threads x 200 blocking calls of 1 024 mixed items each (every scheme, every input format among them), every
    status of every call against the C oracle -- from python threads, and from a C program with pthreads (tests/c/
    thread_client.c: no interpreter lock between the callers, what a service written in the reference's language sees).
    And the threads are served together: with 4 threads the engine completes at least 2 x the calls per second of one thread
    (a launch of the four calls' 4 096 signatures takes 0.55 ms on the device against 0.42 for one call's 1 024, and its host side
    -- four copies in, one upload, the statuses out, the callers coming back -- is in series with it: 2.3-3.2 x observed)."""
    from jubjub_schnorr_amd.tools.small_host_calls import build_thread_client, c_threads, write_batches
    n, calls = 1024, 200
    work = _mixed_work(n)
    for scheme, fmt, args, want in work:            # first calls: buffers grow, lanes are created
        st, tally = _call(eng, scheme, fmt, args)
        assert (st == want).all() and tally.tolist() == [int((want == k).sum()) for k in range(4)]

    bad, start = [], threading.Barrier(5)

    def run(t):
        mine = work[t:] + work[:t]                  # every thread cycles through all six shapes, out of phase with the others
        start.wait()
        for c in range(calls):
            scheme, fmt, args, want = mine[c % len(mine)]
            st, tally = _call(eng, scheme, fmt, args)
            if not ((st == want).all() and tally.tolist() == [int((want == k).sum()) for k in range(4)]):
                bad.append((t, c, scheme, fmt))
    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    start.wait()
    for th in threads:
        th.join()
    assert not bad, bad[:5]

    exe = build_thread_client(str(tmp_path))
    mixed = str(tmp_path / "mixed.bin")
    write_batches(mixed, work)
    rec = c_threads(exe, mixed, [4], calls, rotate=True)[0]              # the same six shapes from four pthreads
    assert rec["mismatches"] == 0 and rec["errors"] == 0, rec
    same = str(tmp_path / "single.bin")
    # the rate: 1 024 single signatures per call (four batches of the same shape, statuses from the oracle), 1 and 4 threads
    singles = [work[0]]
    for i in range(3):
        b = make_batch("single", n, seed=4200 + i, n_keys=16)
        singles.append(("single", "affine", [b[k] for k in ARG_ORDER["single"]], oracle_verify("single", b)))
    write_batches(same, singles)
    # (three attempts, the best one counts: how the threads fall into step is a matter of scheduling, and a box that is busy
    # with something else for a moment says nothing about the engine; observed on idle boxes: 2.3-3.2 x)
    one = four = None
    for _ in range(3):
        a, b = c_threads(exe, same, [1, 4], calls)
        assert a["mismatches"] == 0 and b["mismatches"] == 0 and a["errors"] == 0 and b["errors"] == 0
        if four is None or b["calls_per_s"] / a["calls_per_s"] > four["calls_per_s"] / one["calls_per_s"]:
            one, four = a, b
        if four["calls_per_s"] >= 2.5 * one["calls_per_s"]:
            break
    print(f"calls/s of 1 024 single signatures (C client): 1 thread {one['calls_per_s']:.0f}, 4 threads {four['calls_per_s']:.0f} "
          f"({four['calls_per_s'] / one['calls_per_s']:.2f} x), {four['lane_calls'] / max(1, four['lane_launches']):.2f} calls per launch; "
          f"six shapes in turn, 4 threads: {rec['calls_per_s']:.0f}")
    assert four["calls_per_s"] >= 2.0 * one["calls_per_s"], (one, four)
    assert four["lane_calls"] == 4 * calls and four["lane_launches"] < four["lane_calls"]       # calls shared launches


@pytest.mark.parametrize("n", [1, 64, 4096, 16385, 131072])
def test_lane_calls_of_every_size_class(eng, n):
    """One item, a wave, the largest 8-piece latency call, the first medium size, the largest lane call: statuses of the host
    call equal those of the resident call of the same items (which the parity suite pins to the oracle) and, up to 4 096
    items, the oracle's."""
    import torch
    for scheme in ("single", "double", "vargen"):
        if n > 4096:
            import bench
            arrays, expect = bench.make_inputs(eng, scheme, n, 3, n_keys=max(2, n // 64))
            host = [arrays[k].cpu().numpy() for k in ARG_ORDER[scheme]]
            want = expect.cpu().numpy()
        else:
            b = make_batch(scheme, n, seed=77 + n, n_keys=8)
            host = [b[k] for k in ARG_ORDER[scheme]]
            want = oracle_verify(scheme, b)
        st, tally = eng.verify(scheme, *host)
        assert (st == want).all(), (scheme, n, np.where(st != want)[0][:8])
        assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
        st_d, _ = eng.verify(scheme, *[torch.from_numpy(a).cuda() for a in host])
        assert (st_d.cpu().numpy() == want).all()


def test_host_wire_call_with_unique_keys_after_one_with_repeating_keys(eng):
    """Two host-buffer wire calls in one call slot, back to back: the first leaves the slot's decision word at `key tables,
    wide windows`, the second -- keys that do not repeat -- hashes its first ranges before its own key kernels are queued
    (PREP_ALL reads the word).  Those ranges must find the word cleared, not the first call's decision: a stale read would
    give them records without half-size scalars and without validated keys.  Statuses by construction (GPU signer, pinned to
    the oracle by test_gpu_signer_matches_oracle), invalid keys among the items of the leading ranges."""
    import bench
    import torch
    n = 3 << 17                                  # a large host call: the piece-by-piece pipeline, leading ranges of 2^16 and 2^17 items
    for scheme in ("single", "vargen"):
        rep, want_rep = bench.make_inputs(eng, scheme, n, 11)                    # 4 096 keys: key tables
        uniq, want_uniq = bench.make_inputs(eng, scheme, n, 12, n_keys=n)        # every signature under its own key
        from jubjub_schnorr_amd.tools.small_host_calls import formats_of
        w_rep, w_uniq = formats_of(eng, bench, scheme, rep)["wire"], formats_of(eng, bench, scheme, uniq)["wire"]
        torch.cuda.synchronize()
        before = eng.path_stats()
        for _ in range(3):
            st, tally = eng.verify_wire(scheme, *w_rep)
            assert (st == want_rep.cpu().numpy()).all()
            st, tally = eng.verify_wire(scheme, *w_uniq)
            want = want_uniq.cpu().numpy()
            assert (st == want).all(), np.where(st != want)[0][:8]
            assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
        after = eng.path_stats()
        assert after["key_tables_wide"] + after["key_tables_narrow"] == before["key_tables_wide"] + before["key_tables_narrow"] + 3, (before, after)
        assert after["keys_do_not_repeat"] >= before["keys_do_not_repeat"] + 2, (before, after)      # the last call is counted when its slot is next used


def test_first_large_call_does_not_stall_small_calls_of_another_thread():
    """jjs_reserve pre-sizes the engine; no call waits for the device to grow a buffer.  In a fresh process: a thread verifies
    64 signatures per blocking call in a loop while the main thread issues the FIRST 2^20-item resident call, then a second
    one.  The slowest small call beside the first large call is no slower than the slowest beside the second one plus one
    small-call latency -- the first call did not stop the world to allocate."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def child(*flags):
        p = subprocess.run([sys.executable, os.path.join(root, "tests", "reserve_child.py"), *flags], capture_output=True, text=True,
                           timeout=600, cwd=root)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    rec = child("--reserve")
    print(rec)
    assert rec["bit_exact"] is True
    assert rec["allocated_by_first_call_bytes"] == 0, rec            # everything the call needed had been reserved
    # "one small-call latency" is that of a small call beside a 2^20 batch (several ms: the chip is full), i.e. what the
    # loop sees beside the SECOND large call, which allocates nothing by construction
    assert rec["max_small_ms_beside_first"] <= 2 * rec["max_small_ms_beside_second"] + 0.5, rec
    # the control: without the reservation the same first call allocates inside the call (and still waits for nobody)
    ctl = child()
    print(ctl)
    assert ctl["bit_exact"] is True and ctl["allocated_by_first_call_bytes"] > 1 << 20 and ctl["allocated_by_second_call_bytes"] == 0, ctl


def test_trim_returns_retired_buffers_and_key_pools(eng):
    import bench
    import torch
    arrays, expect = bench.make_inputs(eng, "single", 1 << 17, 5)
    st, _ = eng.verify("single", *[arrays[k] for k in ARG_ORDER["single"]])
    torch.cuda.synchronize()
    assert torch.equal(st, expect)
    before = eng.memory_stats()
    assert before["key_pools"] > 0
    eng.trim()
    after = eng.memory_stats()
    assert after["key_pools"] == 0 and after["retired"] == 0
    st, _ = eng.verify("single", *[arrays[k] for k in ARG_ORDER["single"]])       # the pool comes back
    torch.cuda.synchronize()
    assert torch.equal(st, expect) and eng.memory_stats()["key_pools"] > 0         # (of the slot whose turn it was)


def test_small_calls_proceed_while_a_large_host_call_runs(eng):
    """A blocking host-buffer call of 2^20 items holds its device's host_mu, not the engine's mutex: a second thread's calls of
    256 items start AND end while it runs (before round 4 they waited for it to return), every status right on both sides."""
    import bench
    import torch
    big, big_expect = bench.make_inputs(eng, "single", 1 << 20, 21)
    big_host = [big[k].cpu().numpy() for k in ARG_ORDER["single"]]
    want_big = big_expect.cpu().numpy()
    sm = make_batch("single", 256, seed=9100, n_keys=4)
    small_host = [sm[k] for k in ARG_ORDER["single"]]
    want_small = oracle_verify("single", sm)
    torch.cuda.synchronize()
    eng.verify("single", *big_host); eng.verify("single", *small_host)           # buffers grow
    windows, log, bad, stop = [], [], [], threading.Event()

    def small_loop():
        while not stop.is_set():
            t0 = time.perf_counter()
            st, tally = eng.verify("single", *small_host)
            log.append((t0, time.perf_counter()))
            if not (st == want_small).all():
                bad.append(len(log))
    th = threading.Thread(target=small_loop)
    th.start()
    for _ in range(4):
        t0 = time.perf_counter()
        st, tally = eng.verify("single", *big_host)
        windows.append((t0, time.perf_counter()))
        assert (st == want_big).all() and tally.tolist() == [int((want_big == k).sum()) for k in range(4)]
    stop.set()
    th.join()
    assert not bad
    inside = sum(1 for (a, b) in log for (w0, w1) in windows if a > w0 and b < w1)
    print(f"{len(log)} small calls, {inside} of them began and ended inside one of the 4 large calls "
          f"({(windows[-1][1] - windows[-1][0]) * 1e3:.1f} ms each)")
    assert inside >= 4, (inside, len(log))


def test_lane_state_machine_under_sixteen_threads(eng, tmp_path):
    """Sixteen pthreads, each rotating through twelve batches that differ in scheme, input format and size -- one item, a
    ragged wave, sizes on both sides of the limit up to which calls combine (4 096) and the largest lane call (16 384) -- so
    that lanes fill, close, launch and free in every order: joiners during the gather window, calls too large to combine
    beside combined ones, more callers than lanes.  Every status and tally of every call against the C oracle."""
    from jubjub_schnorr_amd.tools.small_host_calls import build_thread_client, c_threads, write_batches
    specs = [("single", "affine", 1), ("single", "affine", 1000), ("double", "affine", 63), ("vargen", "affine", 4096),
             ("single", "ext", 4097), ("single", "wire", 777), ("double", "wire", 2048), ("vargen", "ext", 300),
             ("single", "affine", 16384), ("double", "ext", 5000), ("vargen", "wire", 1), ("single", "affine", 4096)]
    work = []
    for i, (scheme, fmt, n) in enumerate(specs):
        b = make_batch(scheme, n, seed=7300 + i, n_keys=max(1, min(32, n // 8)))
        want = oracle_verify(scheme, b)
        args = {"affine": [b[k] for k in ARG_ORDER[scheme]], "ext": batch_to_extended(scheme, b, seed=i),
                "wire": list(to_wire(scheme, b))}[fmt]
        work.append((scheme, fmt, [np.ascontiguousarray(a) for a in args], want))
    path = str(tmp_path / "many.bin")
    write_batches(path, work)
    exe = build_thread_client(str(tmp_path))
    rec = c_threads(exe, path, [16], 36, rotate=True)[0]
    print(rec)
    assert rec["mismatches"] == 0 and rec["errors"] == 0, rec
    assert rec["lane_calls"] == 16 * 36 and rec["lane_launches"] <= rec["lane_calls"]


def test_sixty_four_threads_of_one_item_calls(eng, tmp_path):
    """The reference's own API from a busy service: 64 pthreads, ONE signature per blocking call (src/keys/public.rs:114-118),
    valid and invalid ones, both fixed-generator schemes and the per-item-generator one among them; every status against the C
    oracle.  More threads than the box has cores: the members of a lane other than its leader must sleep, not poll -- with
    every waiter polling and woken for every change of every lane (the first version of the lanes) 64 threads completed
    fewer calls per second than 16; now at least 12 x one thread's (observed: 29-35 x)."""
    from jubjub_schnorr_amd.tools.small_host_calls import build_thread_client, c_threads, write_batches
    work = []
    for i in range(64):
        scheme = ("single", "single", "double", "vargen")[i % 4] if i >= 32 else "single"
        b = make_batch(scheme, 1, seed=9100 + i, n_keys=1, mix=False)
        if i % 3 == 1:
            b["m"][0, 0] ^= 1                        # the signature no longer fits the message
        if i % 7 == 3:
            b["PK"][0] = IDENT                       # not a valid key
        work.append((scheme, "affine", [np.ascontiguousarray(b[k]) for k in ARG_ORDER[scheme]], oracle_verify(scheme, b)))
    exe = build_thread_client(str(tmp_path))
    mixed = str(tmp_path / "one_item_mixed.bin")
    write_batches(mixed, work)
    rec = c_threads(exe, mixed, [64], 100, rotate=True)[0]
    assert rec["mismatches"] == 0 and rec["errors"] == 0, rec
    same = str(tmp_path / "one_item_single.bin")
    write_batches(same, work[:32] + work[:32])
    best = None
    for _ in range(3):
        a, b = c_threads(exe, same, [1, 64], 200)
        assert a["mismatches"] == 0 and b["mismatches"] == 0 and a["errors"] == 0 and b["errors"] == 0
        if best is None or b["calls_per_s"] / a["calls_per_s"] > best[1]["calls_per_s"] / best[0]["calls_per_s"]:
            best = (a, b)
        if b["calls_per_s"] >= 20 * a["calls_per_s"]:
            break
    one, many = best
    print(f"one-item calls/s (C client): 1 thread {one['calls_per_s']:.0f}, 64 threads {many['calls_per_s']:.0f} "
          f"({many['calls_per_s'] / one['calls_per_s']:.1f} x), {many['lane_calls'] / max(1, many['lane_launches']):.1f} calls per launch")
    assert many["calls_per_s"] >= 12 * one["calls_per_s"], (one, many)


@pytest.mark.parametrize("n", [0, 1, 100, 5000, 20000])
def test_host_calls_with_optional_outputs(eng, n):
    """The blocking entry points with the outputs a caller may leave out (include/jjs_gpu.h: status and tally are nullable), an
    empty batch, and a null input column: through the lanes (n <= 16 384) and through the pipeline (beyond)."""
    import ctypes
    from jubjub_schnorr_amd import _ffi
    lib = _ffi.lib()
    b = make_batch("single", max(n, 1), seed=6100 + n, n_keys=8)
    want = oracle_verify("single", b)[:n]
    cols = [np.ascontiguousarray(b[k][:n]) for k in ARG_ORDER["single"]]
    ptrs = [c.ctypes.data_as(ctypes.c_void_p) for c in cols]
    st = np.full(n, 0xEE, np.uint8)
    tally = np.full(4, 77, np.uint64)
    p_st, p_tally = st.ctypes.data_as(ctypes.c_void_p), tally.ctypes.data_as(ctypes.c_void_p)
    assert lib.jjs_verify_single(*ptrs, n, p_st, p_tally) == 0
    assert (st == want).all() and tally.tolist() == [int((want == k).sum()) for k in range(4)]
    st[:] = 0xEE
    assert lib.jjs_verify_single(*ptrs, n, p_st, None) == 0              # statuses only
    assert (st == want).all()
    tally[:] = 77
    assert lib.jjs_verify_single(*ptrs, n, None, p_tally) == 0           # tally only
    assert tally.tolist() == [int((want == k).sum()) for k in range(4)]
    assert lib.jjs_verify_single(*ptrs, n, None, None) == 0              # nothing asked for: still a valid call
    if n:
        bad = list(ptrs)
        bad[2] = None
        assert lib.jjs_verify_single(*bad, n, p_st, p_tally) != 0        # a missing column is refused
        assert b"null" in lib.jjs_last_error()
