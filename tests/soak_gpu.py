"""One-off soak run (not collected by pytest): a BASELINE-size batch per scheme, the bench mix plus small-order
components injected into R / R' / Gen, EVERY status compared with the C oracle (about a minute of 16 host
threads per scheme).  Usage on the GPU box: python tests/soak_gpu.py [log2n]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]

import torch  # noqa: E402

import bench  # noqa: E402
import jjs_oracle_c as oc  # noqa: E402
import jubjub_schnorr_amd as jjs  # noqa: E402
from helpers import ARG_ORDER, oracle_verify, pt_arr, torsion_generator  # noqa: E402
import jjs_oracle as o  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = 1 << log2n
    eng = jjs.engine()
    t8 = torsion_generator()
    tors = pt_arr([o.mul(t8, k) for k in range(1, 8)])
    rng = np.random.default_rng(99)
    for scheme in ("single", "double", "vargen"):
        arrays, _ = bench.make_inputs(eng, scheme, n, 0)
        host = {k: v.cpu().numpy().copy() for k, v in arrays.items()}
        # small-order components on 1/64 of the items, spread over the points the bench mix leaves clean
        for name in [k for k in ("R", "Rp", "Gen") if k in host]:
            rows = rng.choice(n, n // 64, replace=False)
            host[name][rows] = oc.point_add(host[name][rows], tors[rng.integers(0, 7, len(rows))])
        t0 = time.time()
        want = oracle_verify(scheme, host)
        t_cpu = time.time() - t0
        st, tally = eng.verify(scheme, *[torch.from_numpy(host[k]).cuda() for k in ARG_ORDER[scheme]])
        st = st.cpu().numpy()
        bad = int((st != want).sum())
        print(f"{scheme}: {n} items, oracle {t_cpu:.0f} s, statuses {np.bincount(want, minlength=4).tolist()}, "
              f"tally {tally.cpu().numpy().tolist()}, mismatches {bad}", flush=True)
        assert bad == 0 and tally.cpu().numpy().tolist() == np.bincount(want, minlength=4).tolist()
        # the same batch through the wire entry points: points compressed on the device, decoded again by the
        # decoder; every point here is on the curve, so the statuses must be the same
        dev = {k: torch.from_numpy(v).cuda() for k, v in host.items()}
        comp = {k: eng.compress(dev[k]) for k in dev if dev[k].shape[1] == 64}
        if scheme == "single":
            sig, pk = torch.cat([dev["u"], comp["R"]], 1), comp["PK"]
        elif scheme == "double":
            sig, pk = torch.cat([dev["u"], comp["R"], comp["Rp"]], 1), torch.cat([comp["PK"], comp["PKp"]], 1)
        else:
            sig, pk = torch.cat([dev["u"], comp["R"]], 1), torch.cat([comp["PK"], comp["Gen"]], 1)
        st_w, tally_w = eng.verify_wire(scheme, sig.contiguous(), pk.contiguous(), dev["m"])
        bad_w = int((st_w.cpu().numpy() != want).sum())
        print(f"{scheme}: wire entry point, mismatches {bad_w}", flush=True)
        assert bad_w == 0 and tally_w.cpu().numpy().tolist() == tally.cpu().numpy().tolist()
    print("SOAK OK")


if __name__ == "__main__":
    main()
