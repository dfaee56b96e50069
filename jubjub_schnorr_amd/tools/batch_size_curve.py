#!/usr/bin/env python3
"""Latency and rate of one device-resident verify call as a function of the batch size: where the GPU path
starts to pay (a lane verifies one signature start to finish, so a small batch costs one signature's latency).
One JSON line per size.  Usage: python jubjub_schnorr_amd/tools/batch_size_curve.py [scheme]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "single"
    eng = jjs.engine()
    arrays, expect = bench.make_inputs(eng, scheme, 1 << 20, 0)
    for n in (1, 64, 256, 1024, 4096, 16384, 65536, 131072, 262144, 1 << 20):
        call = [arrays[k][:n].contiguous() for k in bench.ARG_ORDER[scheme]]
        st, _ = eng.verify(scheme, *call)
        torch.cuda.synchronize()
        assert torch.equal(st, expect[:n])
        times = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.verify(scheme, *call); e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = sorted(times)[len(times) // 2]
        print(json.dumps({"scheme": scheme, "items": n, "ms": ms, "verifications_per_s": n / (ms * 1e-3)}), flush=True)


if __name__ == "__main__":
    main()
