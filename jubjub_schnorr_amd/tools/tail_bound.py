#!/usr/bin/env python3
"""What the stretch of a 2^20 batch that is not arithmetic can be worth: an upper bound by measurement.

The resolve pass runs behind key_verify_kernel over the items whose equation failed (their R still needs its own subgroup
test: InvalidPoint or InvalidSignature).  A batch without such items -- every signature valid, no invalid key -- queues
nothing: resolve_kernel leaves at once.  The difference between the two batches, same box, alternating, is everything that
folding the resolve pass into key_verify_kernel could recover (and more: the mixed batch also does the pass's arithmetic,
1.8 % of the batch's instructions, which no folding removes).

    python -m jubjub_schnorr_amd.tools.tail_bound [scheme] [rounds]      -> JSON lines
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "single"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    import torch
    import bench
    import jubjub_schnorr_amd as jjs
    eng = jjs.engine()
    n = 1 << 20
    mixed, expect = bench.make_inputs(eng, scheme, n, 0)
    valid, _ = bench.make_inputs(eng, scheme, n, 0, mix=False)          # the same keys and signer, nothing spoilt
    calls = {"mixed (bench mix: 1/16 of the items not Ok)": [mixed[k] for k in bench.ARG_ORDER[scheme]],
             "all valid (nothing for the resolve pass)": [valid[k] for k in bench.ARG_ORDER[scheme]]}

    def timed(args, steps=20):
        for _ in range(3):
            st, tally = eng.verify(scheme, *args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st, tally = eng.verify(scheme, *args)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, st, tally
    out = {k: [] for k in calls}
    for _ in range(rounds):
        for k, args in calls.items():
            ms, st, tally = timed(args)
            out[k].append(round(ms, 4))
            if k.startswith("all valid"):
                assert int(tally[0]) == n and not st.any()
            else:
                assert torch.equal(st, expect)
    med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
    keys = list(calls)
    print(json.dumps({"what": "2^20 %s signatures under 4 096 keys, ms per batch, alternating, %d rounds of 20 batches" % (scheme, rounds),
                      "ms": out, "median_ms": med, "difference_ms": round(med[keys[0]] - med[keys[1]], 4),
                      "difference_share": round((med[keys[0]] - med[keys[1]]) / med[keys[0]], 4),
                      "note": "upper bound of what a resolve pass folded into key_verify_kernel could recover (it includes the pass's own arithmetic)"}))


if __name__ == "__main__":
    main()
