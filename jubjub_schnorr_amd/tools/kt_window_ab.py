#!/usr/bin/env python3
"""Key-table path with 5-bit against 6-bit windows on the same box, interleaved (profiling build:
jjs_debug_force_path 0x500 = never the wide windows, 0 = the product's choice, which is 6 bits on SURVEY.md 8(d)'s 256
signatures per key).  One JSON line per (scheme, windows).  Usage: python jubjub_schnorr_amd/tools/kt_window_ab.py"""
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
    _ffi.select_library(_ffi.PROFILING_LIB_PATH)
    eng = jjs.engine()
    lib = _ffi.lib()
    for scheme in ("single", "double", "vargen"):
        arrays, expect = bench.make_inputs(eng, scheme, 1 << 20, 0)
        call = [arrays[k] for k in bench.ARG_ORDER[scheme]]
        res = {5: [], 6: []}
        for rnd in range(3):
            for bits, code in ((5, 0x500), (6, 0)):
                _ffi.check(lib.jjs_debug_force_path(code), "force_path")
                st, _ = eng.verify(scheme, *call)
                torch.cuda.synchronize()
                assert torch.equal(st, expect), (scheme, bits)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.verify(scheme, *call)
                e1.record(); torch.cuda.synchronize()
                res[bits].append(e0.elapsed_time(e1) / 5)
        for bits in (5, 6):
            print(json.dumps({"scheme": scheme, "window_bits": bits, "ms": sorted(res[bits])[1], "ms_all": res[bits]}), flush=True)
    lib.jjs_debug_force_path(0)


if __name__ == "__main__":
    main()
