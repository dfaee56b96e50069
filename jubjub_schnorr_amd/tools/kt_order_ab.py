#!/usr/bin/env python3
"""Key-table path with the items grouped by key (the product) against the caller's order (profiling build,
jjs_debug_force_path 0x1000), same box, interleaved.  One JSON line per (scheme, order).
Usage: python jubjub_schnorr_amd/tools/kt_order_ab.py"""
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
        res = {"caller": [], "by_key": []}
        for rnd in range(4):
            for name, code in (("caller", 0x1000), ("by_key", 0)):
                _ffi.check(lib.jjs_debug_force_path(code), "force_path")
                st, _ = eng.verify(scheme, *call)
                torch.cuda.synchronize()
                assert torch.equal(st, expect), (scheme, name)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    eng.verify(scheme, *call)
                e1.record(); torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 5)
        for name in ("caller", "by_key"):
            v = sorted(res[name][1:])
            print(json.dumps({"scheme": scheme, "item_order": name, "ms": v[len(v) // 2], "ms_all": res[name]}), flush=True)
    lib.jjs_debug_force_path(0)


if __name__ == "__main__":
    main()
