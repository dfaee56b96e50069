#!/usr/bin/env python3
"""Throughput of the multisig batch entry point on synthetic transcripts: B transcripts of n
participants each, built with the library's own fixed-base/var-base kernels is not possible without the
secret-side arithmetic, so the shares are random (the work done is the same whatever the verdict).
Usage: multisig_rate.py [log2_transcripts] [participants]  -> one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    log2b = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    B = 1 << log2b
    N = B * n
    eng = jjs.engine()
    gen = torch.Generator().manual_seed(1)

    def scal(rows, top):
        t = torch.randint(0, 256, (rows, 32), dtype=torch.uint8, generator=gen); t[:, 31] &= top; return t.cuda()
    # valid curve points: public keys / commitments from the signer kernel (R, PK of random secrets)
    _, R, PK = eng.sign("single", scal(N, 0x07), scal(N, 0x07), scal(N, 0x3F))
    _, S, _ = eng.sign("single", scal(N, 0x07), scal(N, 0x07), scal(N, 0x3F))
    z, m = scal(N, 0x07), scal(B, 0x3F)
    offs = np.arange(B + 1, dtype=np.uint32) * n
    eng.multisig_combine(z, PK, R, S, m, offs)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st, *_ = eng.multisig_combine(z, PK, R, S, m, offs); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(json.dumps({"what": "jjs_multisig_combine_dev", "transcripts": B, "participants_each": n, "ms": best,
                      "shares_per_s": N / (best * 1e-3), "transcripts_per_s": B / (best * 1e-3),
                      "invalid_shares": int((st == 4).sum())}))


if __name__ == "__main__":
    main()
