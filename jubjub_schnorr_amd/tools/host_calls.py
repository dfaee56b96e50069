#!/usr/bin/env python3
"""A few blocking host-buffer calls of 2^20 items in a row (scripts/host_timeline.sh traces the last one).
    host_calls.py <single|double|vargen> <affine|ext|wire> [calls] [log2 items]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "single"
    fmt = sys.argv[2] if len(sys.argv) > 2 else "affine"
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    n = 1 << (int(sys.argv[4]) if len(sys.argv) > 4 else 20)
    import torch
    if len(sys.argv) > 5:                                         # a variant build (A/B runs)
        from jubjub_schnorr_amd import _ffi
        _ffi.select_library(os.path.abspath(sys.argv[5]))
    eng = jjs.engine()
    arrays, expect = bench.make_inputs(eng, scheme, n, 0)
    names = bench.ARG_ORDER[scheme]
    if fmt == "affine":
        args = [arrays[k].cpu().numpy() for k in names]
        fn = lambda: eng.verify(scheme, *args)  # noqa: E731
    elif fmt == "ext":
        z = torch.randint(0, 256, (n, 32), dtype=torch.uint8); z[:, 31] &= 0x3F; z[:, 0] |= 1
        z = z.cuda()
        def to_ext(p):
            return torch.cat([eng.debug_fq_mul(p[:, :32].contiguous(), z), eng.debug_fq_mul(p[:, 32:].contiguous(), z), z], 1).contiguous().cpu().numpy()
        args = [to_ext(arrays[k]) if arrays[k].shape[1] == 64 else arrays[k].cpu().numpy() for k in names]
        fn = lambda: eng.verify_ext(scheme, *args)  # noqa: E731
    else:
        c = {k: eng.compress(v) for k, v in arrays.items() if v.shape[1] == 64}
        if scheme == "single":
            w = [torch.cat([arrays["u"], c["R"]], 1), c["PK"], arrays["m"]]
        elif scheme == "double":
            w = [torch.cat([arrays["u"], c["R"], c["Rp"]], 1), torch.cat([c["PK"], c["PKp"]], 1), arrays["m"]]
        else:
            w = [torch.cat([arrays["u"], c["R"]], 1), torch.cat([c["PK"], c["Gen"]], 1), arrays["m"]]
        args = [x.contiguous().cpu().numpy() for x in w]
        fn = lambda: eng.verify_wire(scheme, *args)  # noqa: E731
    torch.cuda.synchronize()
    want = expect.cpu().numpy()
    times = []
    for _ in range(calls):
        t0 = time.perf_counter()
        st, tally = fn()
        times.append(time.perf_counter() - t0)
        assert (st == want).all()
    # the same batch resident in HBM, for reference (same box, same clock); not under the tracer of host_timeline.sh
    res = 0.0
    if not os.environ.get("JJS_HOST_CALLS_ONLY"):
        dev = [arrays[k] for k in names]
        eng.verify(scheme, *dev); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.verify(scheme, *dev)
        torch.cuda.synchronize()
        res = (time.perf_counter() - t0) / 5
    print(json.dumps({"what": f"jjs_verify_{scheme} {fmt} host buffers", "items": n, "lib": os.path.basename(sys.argv[5]) if len(sys.argv) > 5 else "product",
                      "ms": [round(t * 1e3, 3) for t in times], "median_ms": round(sorted(times)[len(times) // 2] * 1e3, 3),
                      "resident_affine_ms": round(res * 1e3, 3)}))


if __name__ == "__main__":
    main()
