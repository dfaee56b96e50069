#!/usr/bin/env python3
"""Aggregate rate of device-resident single-signature calls of n items issued round-robin on k streams: what a caller
with a block's worth of signatures per call gains from issuing the calls on several streams (call slots,
csrc/jjs_gpu.hip).  One JSON line per (n, k).  Usage: python jubjub_schnorr_amd/tools/concurrent_calls.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    eng = jjs.engine()
    arrays, expect = bench.make_inputs(eng, "single", 1 << 20, 0)
    streams = [torch.cuda.Stream() for _ in range(6)]
    for n in (1024, 4096, 16384, 32768, 65536, 131072, 262144, 1048576):
        calls = 12 if n <= 131072 else 6
        batches = [[arrays[k][i * n:(i + 1) * n].contiguous() for k in bench.ARG_ORDER["single"]] for i in range(min(calls, (1 << 20) // n))]
        for k in (1, 2, 3, 6):
            def issue():
                outs = []
                for i in range(calls):
                    with torch.cuda.stream(streams[i % k]):
                        outs.append(eng.verify("single", *batches[i % len(batches)])[0])
                return outs
            outs = issue()
            torch.cuda.synchronize()
            for i, st in enumerate(outs):
                j = i % len(batches)
                assert torch.equal(st, expect[j * n:(j + 1) * n]), (n, k, i)
            times = []
            for _ in range(5):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s_ in streams[:k]:
                    s_.wait_event(e0)
                issue()
                for s_ in streams[:k]:
                    torch.cuda.current_stream().wait_stream(s_)
                e1.record(); torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            ms = sorted(times)[len(times) // 2]
            print(json.dumps({"items_per_call": n, "streams": k, "calls": calls, "ms": ms, "ms_per_call": ms / calls,
                              "verifications_per_s": calls * n / (ms * 1e-3)}), flush=True)


if __name__ == "__main__":
    main()
