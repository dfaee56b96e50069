#!/usr/bin/env python3
"""Latency and rate of one device-resident verify call as a function of the batch size, for both paths of the
engine: the throughput path (a lane verifies one signature start to finish, so a call costs at least one
signature's latency) and the latency path (csrc/small_batch.h: a signature spread over 11 / 21 lanes).  Loads the
profiling build, whose jjs_debug_force_path selects the path regardless of size; "auto" is what the product does.
One JSON line per (size, path).  Usage: python jubjub_schnorr_amd/tools/batch_size_curve.py [scheme]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402
from jubjub_schnorr_amd import _ffi  # noqa: E402


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "single"
    _ffi.select_library(_ffi.PROFILING_LIB_PATH)
    eng = jjs.engine()
    lib = _ffi.lib()
    arrays, expect = bench.make_inputs(eng, scheme, 1 << 20, 0)
    sizes = (1, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20)
    for n in sizes:
        call = [arrays[k][:n].contiguous() for k in bench.ARG_ORDER[scheme]]
        for path, code in (("auto", 0), ("throughput", 1), ("latency-4", 0x42), ("latency-8", 0x82)):
            if path.startswith("latency") and n > 65536:
                continue
            _ffi.check(lib.jjs_debug_force_path(code), "force_path")
            st, _ = eng.verify(scheme, *call)
            torch.cuda.synchronize()
            assert torch.equal(st, expect[:n]), (n, path)
            times = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); eng.verify(scheme, *call); e1.record(); torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ms = sorted(times)[len(times) // 2]
            print(json.dumps({"scheme": scheme, "items": n, "path": path, "ms": ms, "ms_min": min(times),
                              "verifications_per_s": n / (ms * 1e-3)}), flush=True)
    lib.jjs_debug_force_path(0)


if __name__ == "__main__":
    main()
