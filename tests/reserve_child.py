"""Child process of tests/test_host_threads_gpu.py: a fresh engine, a thread of small blocking host-buffer calls, and beside
it the FIRST 2^20-item resident call of the process, then a second one.  `--reserve`: jjs_reserve first.  One JSON line."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reserve = "--reserve" in sys.argv
    import torch
    import bench
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    n_big, n_small = 1 << 20, 64
    if reserve:
        eng.reserve("single", n_big)
        eng.reserve("single", n_small, host_buffers=True)
    arrays, expect = bench.make_inputs(eng, "single", n_big, 0)
    big = [arrays[k] for k in bench.ARG_ORDER["single"]]
    sm, sm_expect = bench.make_inputs(eng, "single", n_small, 1, n_keys=4)
    small = [sm[k].cpu().numpy() for k in bench.ARG_ORDER["single"]]
    want_small = sm_expect.cpu().numpy()
    for _ in range(5):
        eng.verify("single", *small)
    alone = []
    for _ in range(200):
        t0 = time.perf_counter()
        eng.verify("single", *small)
        alone.append(time.perf_counter() - t0)
    total = lambda: sum(eng.memory_stats().values())      # noqa: E731
    log, stop, bad = [], threading.Event(), []

    def loop():
        while not stop.is_set():
            t0 = time.perf_counter()
            st, _ = eng.verify("single", *small)
            log.append((t0, time.perf_counter() - t0))
            if not (st == want_small).all():
                bad.append(len(log))
    th = threading.Thread(target=loop)
    th.start()
    time.sleep(0.05)
    stream = torch.cuda.Stream()
    windows, statuses, mem = [], [], [total()]
    for _ in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            st, _ = eng.verify("single", *big)
        stream.synchronize()
        windows.append((t0, time.perf_counter()))
        statuses.append(st)
        mem.append(total())
        time.sleep(0.05)
    stop.set()
    th.join()
    ok = all(torch.equal(s, expect) for s in statuses) and not bad

    def worst(w):
        hit = [d for (t, d) in log if t < w[1] and t + d > w[0]]
        return max(hit) * 1e3 if hit else 0.0
    print(json.dumps({"reserved": reserve, "bit_exact": bool(ok), "median_small_ms_alone": sorted(alone)[len(alone) // 2] * 1e3,
                      "max_small_ms_beside_first": worst(windows[0]), "max_small_ms_beside_second": worst(windows[1]),
                      "first_big_call_ms": (windows[0][1] - windows[0][0]) * 1e3, "second_big_call_ms": (windows[1][1] - windows[1][0]) * 1e3,
                      "allocated_by_first_call_bytes": mem[1] - mem[0], "allocated_by_second_call_bytes": mem[2] - mem[1],
                      "small_calls_logged": len(log)}), flush=True)


if __name__ == "__main__":
    main()
