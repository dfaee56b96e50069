"""Soak: a BASELINE-size batch per scheme -- the bench mix plus small-order components injected into R / R' / Gen --
with EVERY status compared with the C oracle (the reference's algorithm on all host cores), then the same batch
compressed on the device and pushed through the wire entry points, and then through the blocking host-buffer entry
points in all three formats (affine, extended U || V || Z, wire: the piece-by-piece upload pipeline of run_host_block).  Collected under `-m gpu` by
tests/test_soak_gpu.py; also runnable by hand on the GPU box:  python tests/soak_gpu.py [log2n] [out.json]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]


def run_soak(log2n=20, schemes=("single", "double", "vargen")) -> dict:
    """log2n: one size for every scheme, or a dict {scheme: log2 of its item count}."""
    import torch

    import bench
    import jjs_oracle as o
    import jjs_oracle_c as oc
    import jubjub_schnorr_amd as jjs
    from helpers import ARG_ORDER, pt_arr, torsion_generator

    try:
        oc.build(native=True)          # the GPU box's own CPU: ~15 % faster than the generic build
        native = True
    except Exception:
        native = False
    fns = {"single": oc.verify_single, "double": oc.verify_double, "vargen": oc.verify_vargen}
    sizes = {s: 1 << (log2n[s] if isinstance(log2n, dict) else log2n) for s in schemes}
    eng = jjs.engine()
    t8 = torsion_generator()
    tors = pt_arr([o.mul(t8, k) for k in range(1, 8)])
    rng = np.random.default_rng(99)
    info = bench.host_info()
    probe = {k: v[:4096] for k, v in bench.make_inputs(eng, "single", 4096, 0)[0].items()}
    probe = [probe[k].cpu().numpy() for k in ARG_ORDER["single"]]

    def probe_rate(t):
        t0 = time.time()
        oc.verify_single(*probe, threads=t, native=native)
        return 4096 / (time.time() - t0)

    threads, _, _ = bench.pick_threads(probe_rate, info, oc.max_threads(native))
    report = {"items_per_scheme": sizes, "csrc_sha256": bench.csrc_hash(), "oracle": "oracle/jjs_oracle.c", "host": info,
              "oracle_threads": threads, "schemes": {}}
    for scheme in schemes:
        n = sizes[scheme]
        arrays, _ = bench.make_inputs(eng, scheme, n, 0)
        host = {k: v.cpu().numpy().copy() for k, v in arrays.items()}
        # small-order components on 1/64 of the items, spread over the points the bench mix leaves clean
        for name in [k for k in ("R", "Rp", "Gen") if k in host]:
            rows = rng.choice(n, n // 64, replace=False)
            host[name][rows] = oc.point_add(host[name][rows], tors[rng.integers(0, 7, len(rows))])
        t0 = time.time()
        want = fns[scheme](*[host[k] for k in ARG_ORDER[scheme]], threads=threads, native=native)
        t_cpu = time.time() - t0
        st, tally = eng.verify(scheme, *[torch.from_numpy(host[k]).cuda() for k in ARG_ORDER[scheme]])
        st = st.cpu().numpy()
        bad = int((st != want).sum())
        hist = np.bincount(want, minlength=4).tolist()
        print(f"{scheme}: {n} items, oracle {t_cpu:.0f} s on {threads} threads, statuses {hist}, "
              f"tally {tally.cpu().numpy().tolist()}, mismatches {bad}", flush=True)
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
        # the blocking host-buffer entry points on the same statuses: pageable arrays in, statuses out
        from helpers import ext_on_device
        names = ARG_ORDER[scheme]
        st_h, tally_h = eng.verify(scheme, *[host[k] for k in names])
        ext = [ext_on_device(eng, host[k], seed=5) if host[k].shape[1] == 64 else host[k] for k in names]
        st_e, tally_e = eng.verify_ext(scheme, *ext)
        st_ed, tally_ed = eng.verify_ext(scheme, *[torch.from_numpy(a).cuda() for a in ext])      # resident: normalised beside the hashes
        bad_ext_dev = int((st_ed.cpu().numpy() != want).sum()) + (0 if tally_ed.cpu().numpy().tolist() == hist else 1)
        st_hw, tally_hw = eng.verify_wire(scheme, sig.contiguous().cpu().numpy(), pk.contiguous().cpu().numpy(), host["m"])
        bad_host = {"affine": int((st_h != want).sum()), "ext": int((st_e != want).sum()), "wire": int((st_hw != want).sum()),
                    "ext_resident": bad_ext_dev}
        tallies_host = [t.tolist() for t in (tally_h, tally_e, tally_hw)]
        print(f"{scheme}: host-buffer entry points, mismatches {bad_host}", flush=True)
        del ext
        report["schemes"][scheme] = {"oracle_status_histogram": hist, "gpu_tally": tally.cpu().numpy().tolist(),
                                     "gpu_tally_wire": tally_w.cpu().numpy().tolist(), "mismatches_affine": bad,
                                     "mismatches_wire": bad_w, "mismatches_host_buffers": bad_host,
                                     "host_buffer_tallies_equal": all(t == hist for t in tallies_host),
                                     "oracle_seconds": round(t_cpu, 1)}
        del dev, comp, arrays
        torch.cuda.empty_cache()
    report["ok"] = all(r["mismatches_affine"] == 0 and r["mismatches_wire"] == 0 and not any(r["mismatches_host_buffers"].values()) and
                       r["host_buffer_tallies_equal"] and
                       r["gpu_tally"] == r["oracle_status_histogram"] == r["gpu_tally_wire"] for r in report["schemes"].values())
    return report


def write_report(report: dict, path: str) -> None:
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump(report, open(path, "w"), indent=1)
    except OSError:
        pass


if __name__ == "__main__":
    rep = run_soak(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
    write_report(rep, sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "..", "gpurun_out", "soak.json"))
    print("SOAK OK" if rep["ok"] else "SOAK FAILED")
    sys.exit(0 if rep["ok"] else 1)
