"""Where a combined launch of small host-buffer calls spends its time: runs tests/c/thread_client.c against a build with
-DJJS_LANE_TRACE (libjjs_gpu_trace.so: hipcc ... -DJJS_LANE_TRACE) for one thread count at a time and prints the engine's sums
(open -> ready to launch -> queued -> device done -> last member gone) beside the client's rate.
    python scripts/lane_trace.py [items_per_call] [threads,threads,...] [lib]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,16").split(",")]
    lib = os.path.abspath(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "jubjub_schnorr_amd", "libjjs_gpu_trace.so"))
    import bench
    import jubjub_schnorr_amd as jjs
    from jubjub_schnorr_amd.tools import small_host_calls as shc
    eng = jjs.engine()
    tmp = tempfile.mkdtemp(prefix="jjs_trace_")
    batches = []
    for t in range(max(counts)):
        arrays, expect = bench.make_inputs(eng, "single", n, 100 + t, n_keys=max(2, n // 16))
        batches.append(("single", "affine", shc.formats_of(eng, bench, "single", arrays, seed=7 + t)["affine"], expect.cpu().numpy()))
    path = os.path.join(tmp, "batches.bin")
    shc.write_batches(path, batches)
    exe = shc.build_thread_client(tmp, lib)
    for t in counts:
        p = subprocess.run([exe, path, str(t), "200"], capture_output=True, text=True, timeout=300)
        rec = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        tr = [json.loads(l) for l in p.stderr.splitlines() if l.startswith('{"lane_trace"')]
        print(json.dumps({"items_per_call": n, "threads": t, "calls_per_s": rec[0]["calls_per_s"] if rec else None,
                          "lane_trace": tr[-1]["lane_trace"] if tr else None, "lib": os.path.basename(lib)}), flush=True)


if __name__ == "__main__":
    main()
