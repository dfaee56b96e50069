"""Round start against round end on ONE box: the blocking host-buffer entry points from tests/c/thread_client.c (a C program:
it needs nothing of the library but the verify entry points, so it links against an older build as well), one thread (the
latency of a call) and several, the product build and a second build alternating.
    python scripts/round_ab.py <other lib> [rounds]        e.g. jubjub_schnorr_amd/libjjs_gpu_r03.so = round 3's final code"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    other = os.path.abspath(sys.argv[1])
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    import bench
    import jubjub_schnorr_amd as jjs
    from jubjub_schnorr_amd.tools import small_host_calls as shc
    eng = jjs.engine()
    tmp = tempfile.mkdtemp(prefix="jjs_round_ab_")
    cases = [("single", "affine", 1, (1, 4, 8, 16, 64)), ("single", "affine", 64, (1,)), ("single", "affine", 1024, (1, 4, 8)),
             ("single", "affine", 4096, (1,)), ("single", "affine", 16384, (1,)), ("single", "ext", 64, (1,)), ("single", "wire", 64, (1,)),
             ("double", "affine", 64, (1,)), ("double", "wire", 64, (1,)), ("vargen", "affine", 64, (1,)), ("vargen", "affine", 1024, (1,))]
    files = []
    for scheme, fmt, n, counts in cases:
        batches = []
        for t in range(max(counts)):
            arrays, expect = bench.make_inputs(eng, scheme, n, 100 + t, n_keys=max(2, n // 16))
            batches.append((scheme, fmt, shc.formats_of(eng, bench, scheme, arrays, seed=7 + t)[fmt], expect.cpu().numpy()))
        path = os.path.join(tmp, "b_%s_%s_%d.bin" % (scheme, fmt, n))
        shc.write_batches(path, batches)
        files.append(path)
    exes = {"round end": shc.build_thread_client(tmp), os.path.basename(other): shc.build_thread_client(tmp, other)}
    for rnd in range(rounds):
        for (scheme, fmt, n, counts), path in zip(cases, files):
            for name, exe in exes.items():
                recs = shc.c_threads(exe, path, counts, 200 if n <= 4096 else 60)
                print(json.dumps({"lib": name, "scheme": scheme, "format": fmt, "items_per_call": n,
                                  "calls_per_s": {str(r["threads"]): round(r["calls_per_s"]) for r in recs},
                                  "ms_per_call_one_thread": round(1e3 / recs[0]["calls_per_s"], 4), "mismatches": sum(r["mismatches"] for r in recs)}), flush=True)


if __name__ == "__main__":
    main()
