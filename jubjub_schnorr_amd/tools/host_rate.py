#!/usr/bin/env python3
"""PCIe-inclusive rate of the blocking host-buffer entry point (never bench.py's `value`): numpy arrays in,
status out, including hipMalloc/H2D/D2H inside jjs_verify_single.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    scheme = os.environ.get("JJS_HOST_RATE_SCHEME", "single")     # single (default, also times the wire entry point), double, vargen
    timing = len(sys.argv) > 2 and sys.argv[2] == "timing"      # profiling build: where the host time of a call goes
    from jubjub_schnorr_amd import _ffi
    if timing:
        _ffi.select_library(_ffi.PROFILING_LIB_PATH)
    if len(sys.argv) > 3:                                         # a variant build (A/B runs)
        _ffi.select_library(os.path.abspath(sys.argv[3]))
    eng = jjs.engine()
    arrays, expect = bench.make_inputs(eng, scheme, 1 << log2n, 0)
    host = [arrays[k].cpu().numpy() for k in bench.ARG_ORDER[scheme]]
    eng.verify(scheme, *host)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        st, tally = eng.verify(scheme, *host)
        best = min(best, time.perf_counter() - t0)
    assert (st == expect.cpu().numpy()).all()
    rec = {"what": f"jjs_verify_{scheme} host buffers (pageable), PCIe inclusive", "items": 1 << log2n,
           "seconds": best, "verifications_per_s": (1 << log2n) / best}
    if timing:
        import ctypes
        t = (ctypes.c_double * 8)()
        _ffi.check(_ffi.lib().jjs_debug_host_timing(t), "host_timing")
        rec["last_call_ms"] = {"waiting_for_staging": round(t[0] * 1e3, 3), "starting_next_staging": round(t[1] * 1e3, 3),
                               "total": round(t[2] * 1e3, 3), "pieces": int(t[3]), "until_first_upload_queued": round(t[4] * 1e3, 3),
                               "until_all_queued": round(t[5] * 1e3, 3), "draining": round(t[6] * 1e3, 3), "copy_out": round(t[7] * 1e3, 3)}
    print(json.dumps(rec), flush=True)
    if scheme != "single":
        return
    # the same batch as the reference serialises it (64-byte signatures, 32-byte keys): 128 instead of 196 bytes per item
    # over PCIe; on the device one square root per signature (R) and one per distinct key
    import torch
    sig = torch.cat([arrays["u"], eng.compress(arrays["R"])], 1).contiguous().cpu().numpy()
    pk = eng.compress(arrays["PK"]).cpu().numpy()
    wire = [sig, pk, host[3]]
    eng.verify_wire("single", *wire)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        st, tally = eng.verify_wire("single", *wire)
        best = min(best, time.perf_counter() - t0)
    assert (st == expect.cpu().numpy()).all()
    print(json.dumps({"what": "jjs_verify_single_wire host buffers (pageable), PCIe inclusive", "items": 1 << log2n,
                      "seconds": best, "verifications_per_s": (1 << log2n) / best}))


if __name__ == "__main__":
    main()
