#!/usr/bin/env python3
"""One launch of challenge_kernel (the challenge hash alone: no point checks, no descriptor kept live, 0 spilled SGPRs)
and one verification of the same 2^20 single-signature batch (prepare_kernel: the same hash inside the verification
kernel, 180 spilled SGPRs).  Run under  rocprofv3 --pmc SQ_INSTS_VALU  by scripts/sgpr_spill_cost.sh, which compares
the instruction counts of the two kernels: what the spill traffic (v_readlane / v_writelane) in the hash loop of
prepare_kernel can cost at most."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402


def main():
    import torch
    eng = jjs.engine()
    arrays, expect = bench.make_inputs(eng, "single", 1 << 20, 0)
    eng.challenge("single", arrays["R"], arrays["PK"], arrays["m"])
    st, _ = eng.verify("single", *[arrays[k] for k in bench.ARG_ORDER["single"]])
    torch.cuda.synchronize()
    assert torch.equal(st, expect)


if __name__ == "__main__":
    main()
