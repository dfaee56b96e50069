#!/usr/bin/env python3
"""The reference's own call pattern -- few signatures per call, from host memory, from several threads (the reference
API verifies ONE item per call: src/keys/public.rs:114-118) -- through the blocking host-buffer entry points.

    python -m jubjub_schnorr_amd.tools.small_host_calls [--schemes single,double,vargen] [--formats affine,ext,wire]
                                                         [--sizes 1,64,1024,4096,16384] [--threads 1,2,3,4,6] [--lib PATH]

Two records per line of output (JSON lines):
  * latency: one thread, `jjs_verify_<scheme>{,_ext,_wire}` on pageable numpy arrays of n items, wall time per call
    (median of `reps` calls after two untimed ones), beside the resident `_dev` call of the same items (HIP events);
  * threads: T host threads, each with its own batch of `n_thread` items, `calls` calls per thread; calls/s and
    items/s against the one-thread rate.  Every status is compared with the by-construction expectation.
bench.py quotes `measure()` under `small_host_calls` (a secondary record, never `value`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def formats_of(eng, bench, scheme, arrays, seed=5):
    """numpy argument lists of one batch in the three input formats"""
    import torch
    names = bench.ARG_ORDER[scheme]
    n = arrays["u"].shape[0]
    zgen = torch.Generator(device="cpu").manual_seed(seed)

    def to_ext(pts):
        z = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=zgen)
        z[:, 31] &= 0x3F; z[:, 0] |= 1
        z = z.cuda()
        U = eng.debug_fq_mul(pts[:, :32].contiguous(), z)
        V = eng.debug_fq_mul(pts[:, 32:].contiguous(), z)
        return torch.cat([U, V, z], 1).contiguous().cpu().numpy()
    c = {k: eng.compress(v) for k, v in arrays.items() if v.shape[1] == 64}
    if scheme == "single":
        wire = [torch.cat([arrays["u"], c["R"]], 1), c["PK"], arrays["m"]]
    elif scheme == "double":
        wire = [torch.cat([arrays["u"], c["R"], c["Rp"]], 1), torch.cat([c["PK"], c["PKp"]], 1), arrays["m"]]
    else:
        wire = [torch.cat([arrays["u"], c["R"]], 1), torch.cat([c["PK"], c["Gen"]], 1), arrays["m"]]
    return {"affine": [arrays[k].cpu().numpy() for k in names],
            "ext": [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k].cpu().numpy() for k in names],
            "wire": [w.contiguous().cpu().numpy() for w in wire]}


def call(eng, scheme, fmt, args):
    if fmt == "affine":
        return eng.verify(scheme, *args)
    return eng.verify_ext(scheme, *args) if fmt == "ext" else eng.verify_wire(scheme, *args)


def latency(eng, bench, scheme, sizes, formats, reps=15):
    """ms per blocking host-buffer call (one thread), per format and size; `dev_ms`: the resident call of the same items"""
    import torch
    out = {f: {} for f in formats}
    out["dev_ms"] = {}
    ok = True
    for n in sizes:
        arrays, expect = bench.make_inputs(eng, scheme, n, 0, n_keys=max(2, min(4096, n // 16 or 2)))
        want = expect.cpu().numpy()
        host = formats_of(eng, bench, scheme, arrays)
        for f in formats:
            for _ in range(2):
                call(eng, scheme, f, host[f])
            times = []
            for _ in range(reps):
                t0 = time.perf_counter()
                st, tally = call(eng, scheme, f, host[f])
                times.append(time.perf_counter() - t0)
            ok = ok and bool((st == want).all()) and tally.tolist() == [int((want == k).sum()) for k in range(4)]
            out[f][str(n)] = round(sorted(times)[len(times) // 2] * 1e3, 4)
        dev = [arrays[k] for k in bench.ARG_ORDER[scheme]]
        for _ in range(2):
            eng.verify(scheme, *dev)
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.verify(scheme, *dev); e1.record()
            torch.cuda.synchronize()
            evs.append(e0.elapsed_time(e1))
        out["dev_ms"][str(n)] = round(sorted(evs)[len(evs) // 2], 4)
    return out, ok


def threaded(eng, bench, scheme, fmt, n_thread, thread_counts, calls=200):
    """T threads x `calls` blocking calls of n_thread items each; returns calls/s per T and whether every status was right"""
    most = max(thread_counts)
    work = []
    for t in range(most):
        arrays, expect = bench.make_inputs(eng, scheme, n_thread, 100 + t, n_keys=max(2, n_thread // 16))
        work.append((formats_of(eng, bench, scheme, arrays, seed=7 + t)[fmt], expect.cpu().numpy()))
    for args, _ in work:
        call(eng, scheme, fmt, args)
    rates, ok = {}, True
    for T in thread_counts:
        bad = []
        start = threading.Barrier(T + 1)

        def run(t):
            args, want = work[t]
            start.wait()
            for _ in range(calls):
                st, tally = call(eng, scheme, fmt, args)
                if not ((st == want).all() and tally.tolist() == [int((want == k).sum()) for k in range(4)]):
                    bad.append(t)
        threads = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        start.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        dt = time.perf_counter() - t0
        rates[str(T)] = round(T * calls / dt, 1)
        ok = ok and not bad
    return rates, ok


def threaded_dev(eng, bench, scheme, n_thread, thread_counts, calls=200):
    """The same from resident inputs: T threads, each with a stream of its own, `calls` asynchronous calls each followed by
    a wait for its stream -- what the lanes would reach without their copies."""
    import torch
    most = max(thread_counts)
    work = []
    for t in range(most):
        arrays, expect = bench.make_inputs(eng, scheme, n_thread, 100 + t, n_keys=max(2, n_thread // 16))
        work.append(([arrays[k] for k in bench.ARG_ORDER[scheme]], expect))
    rates, ok = {}, True
    for T in thread_counts:
        bad = []
        start = threading.Barrier(T + 1)

        def run(t):
            args, want = work[t]
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(calls):
                    st, tally = eng.verify(scheme, *args)
                    stream.synchronize()
                if not torch.equal(st, want):
                    bad.append(t)
        threads = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        start.wait()
        t0 = time.perf_counter()
        for th in threads:
            th.join()
        rates[str(T)] = round(T * calls / (time.perf_counter() - t0), 1)
        ok = ok and not bad
    return rates, ok


SCHEME_IDS = {"single": 0, "double": 1, "vargen": 2}
FORMAT_IDS = {"affine": 0, "ext": 1, "wire": 2}


def write_batches(path, batches):
    """The file tests/c/thread_client.c reads: batches = [(scheme, format, [numpy columns], expected statuses)]."""
    import struct
    import numpy as np
    with open(path, "wb") as f:
        f.write(b"JJSB" + struct.pack("<I", len(batches)))
        for scheme, fmt, cols, want in batches:
            n = len(want)
            f.write(struct.pack("<4I", SCHEME_IDS[scheme], FORMAT_IDS[fmt], n, len(cols)))
            f.write(struct.pack("<%dI" % len(cols), *[c.shape[1] for c in cols]))
            for c in cols:
                f.write(np.ascontiguousarray(c, dtype=np.uint8).tobytes())
            f.write(np.ascontiguousarray(want, dtype=np.uint8).tobytes())


def build_thread_client(out_dir, lib=None):
    """gcc tests/c/thread_client.c against the engine (or a variant build of it); returns the executable."""
    import subprocess
    pkg = os.path.join(ROOT, "jubjub_schnorr_amd")
    name = os.path.basename(lib) if lib else "libjjs_gpu.so"
    exe = os.path.join(out_dir, "thread_client_" + name.replace(".so", ""))
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "thread_client.c"), "-o", exe, "-L" + pkg, "-l:" + name,
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def c_threads(exe, batch_file, thread_counts, calls, rotate=False):
    """Runs the C client; returns its records (one per thread count)."""
    import subprocess
    cmd = [exe, batch_file, ",".join(str(t) for t in thread_counts), str(calls)] + (["rotate"] if rotate else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0:
        raise RuntimeError("thread_client failed (%d): %s %s" % (p.returncode, p.stdout[-500:], p.stderr[-1500:]))
    return recs


def measure(eng, bench, schemes=("single",), formats=("affine", "ext", "wire"), sizes=(1, 64, 1024, 4096, 16384),
            thread_counts=(1, 4, 8), n_thread=1024, calls=200, one_item_threads=(1, 8, 64)):
    """What bench.py quotes: per scheme the latency table (one python thread), and for the first scheme the thread scaling of
    the affine calls from the C client (pthreads): calls of n_thread items, and calls of ONE item -- the reference's own API
    (one signature per call, src/keys/public.rs:114-118) from a service's threads."""
    import tempfile
    rec, ok = {"unit": "ms per blocking call, pageable host arrays in, statuses out", "ms_per_call": {}}, True
    for s in schemes:
        lat, good = latency(eng, bench, s, sizes, formats)
        rec["ms_per_call"][s] = lat
        ok = ok and good
    s = schemes[0]
    tmp = tempfile.mkdtemp(prefix="jjs_threads_")
    exe = build_thread_client(tmp)
    for key, n_call, counts in (("threads", n_thread, thread_counts), ("threads_one_item_calls", 1, one_item_threads)):
        if not counts:
            continue
        batches = []
        for t in range(max(counts)):
            arrays, expect = bench.make_inputs(eng, s, n_call, 100 + t, n_keys=max(2, n_call // 16))
            batches.append((s, "affine", formats_of(eng, bench, s, arrays, seed=7 + t)["affine"], expect.cpu().numpy()))
        path = os.path.join(tmp, "batches_%d.bin" % n_call)
        write_batches(path, batches)
        recs = c_threads(exe, path, counts, calls)
        one = recs[0]["calls_per_s"]
        rec[key] = {"scheme": s, "format": "affine", "items_per_call": n_call, "calls_per_thread": calls, "client": "tests/c/thread_client.c (pthreads)",
                    "calls_per_s": {str(r["threads"]): round(r["calls_per_s"]) for r in recs},
                    "speedup_over_one_thread": {str(r["threads"]): round(r["calls_per_s"] / one, 2) for r in recs},
                    "calls_per_launch": {str(r["threads"]): round(r["lane_calls"] / max(1, r["lane_launches"]), 2) for r in recs}}
        ok = ok and all(r["mismatches"] == 0 and r["errors"] == 0 for r in recs)
    rec["bit_exact"] = bool(ok)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schemes", default="single,double,vargen")
    ap.add_argument("--formats", default="affine,ext,wire")
    ap.add_argument("--sizes", default="1,64,1024,4096,16384,65536,131072")
    ap.add_argument("--threads", default="1,2,3,4,6,8")
    ap.add_argument("--items-per-call", type=int, default=1024)
    ap.add_argument("--calls", type=int, default=200)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--dev", action="store_true", help="also the thread scaling of resident (_dev) calls")
    ap.add_argument("--c-client", action="store_true", help="thread scaling from tests/c/thread_client.c (pthreads) instead of python threads")
    args = ap.parse_args()
    import bench
    from jubjub_schnorr_amd import _ffi
    if args.lib:
        _ffi.select_library(os.path.abspath(args.lib))
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    sizes = [int(x) for x in args.sizes.split(",")]
    formats = args.formats.split(",")
    for s in args.schemes.split(","):
        lat, ok = latency(eng, bench, s, sizes, formats)
        print(json.dumps({"what": "latency", "scheme": s, "ms_per_call": lat, "bit_exact": ok, "lib": os.path.basename(_ffi.LIB_PATH)}), flush=True)
    tc = [int(x) for x in args.threads.split(",")]
    if args.c_client:
        # the same thread scaling from a C program (pthreads): no interpreter lock between the callers
        import tempfile
        tmp = tempfile.mkdtemp(prefix="jjs_threads_")
        exe = build_thread_client(tmp, args.lib)
        for s in args.schemes.split(","):
            for f in formats:
                batches = []
                for t in range(max(tc)):
                    arrays, expect = bench.make_inputs(eng, s, args.items_per_call, 100 + t, n_keys=max(2, args.items_per_call // 16))
                    batches.append((s, f, formats_of(eng, bench, s, arrays, seed=7 + t)[f], expect.cpu().numpy()))
                path = os.path.join(tmp, "batches_%s_%s.bin" % (s, f))
                write_batches(path, batches)
                recs = c_threads(exe, path, tc, args.calls)
                one = recs[0]["calls_per_s"]
                print(json.dumps({"what": "threads (C client, pthreads)", "scheme": s, "format": f, "items_per_call": args.items_per_call,
                                  "calls_per_thread": args.calls, "calls_per_s": {str(r["threads"]): r["calls_per_s"] for r in recs},
                                  "speedup": {str(r["threads"]): round(r["calls_per_s"] / one, 2) for r in recs},
                                  "calls_per_launch": {str(r["threads"]): round(r["lane_calls"] / max(1, r["lane_launches"]), 2) for r in recs},
                                  "mismatches": sum(r["mismatches"] for r in recs), "lib": os.path.basename(args.lib or "libjjs_gpu.so")}), flush=True)
        return
    if args.dev:
        for rep in range(2):
            rates, ok = threaded_dev(eng, bench, "single", args.items_per_call, tc, args.calls)
            print(json.dumps({"what": "threads, resident inputs (_dev call + stream wait per call)", "scheme": "single", "items_per_call": args.items_per_call,
                              "calls_per_s": rates, "bit_exact": ok, "lib": os.path.basename(_ffi.LIB_PATH)}), flush=True)
    for s in args.schemes.split(","):
        for f in formats:
            rates, ok = threaded(eng, bench, s, f, args.items_per_call, tc, args.calls)
            one = rates[str(tc[0])]
            print(json.dumps({"what": "threads", "scheme": s, "format": f, "items_per_call": args.items_per_call, "calls_per_thread": args.calls,
                              "calls_per_s": rates, "speedup": {k: round(v / one, 2) for k, v in rates.items()}, "bit_exact": ok,
                              "lib": os.path.basename(_ffi.LIB_PATH)}), flush=True)


if __name__ == "__main__":
    main()
