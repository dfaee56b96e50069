#!/usr/bin/env python3
"""Kernel durations of one small call under rocprofv3 --kernel-trace: small_a_kernel (hash | chains | point checks side by
side) and small_b_kernel, with the full call and with the challenge hash skipped (profiling build), i.e. which of phase A's
roles the call waits for.  Usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/small_call_trace.py <scheme> <n> [skip] [lib]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from jubjub_schnorr_amd import _ffi  # noqa: E402

scheme, n = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(sys.argv) > 4:                                # a variant build (A/B runs)
    _ffi.select_library(os.path.abspath(sys.argv[4]))
elif skip:
    _ffi.select_library(_ffi.PROFILING_LIB_PATH)
import jubjub_schnorr_amd as jjs  # noqa: E402
eng = jjs.engine()
if skip:
    _ffi.check(_ffi.lib().jjs_debug_skip_phases(skip), "skip")
arrays, _ = bench.make_inputs(eng, scheme, n, 0, n_keys=max(2, n // 16))
args = [arrays[k] for k in bench.ARG_ORDER[scheme]]
for _ in range(30):
    eng.verify(scheme, *args)
    torch.cuda.synchronize()
