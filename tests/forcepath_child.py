"""Child process of test_gpu_parity.py::test_both_paths_at_every_size: loads the profiling build and runs the
same ragged batches through the throughput path and through the latency path (jjs_debug_force_path), every status
against the oracle.  The product library chooses between the two by size only; this pins each of them at sizes
the other normally serves."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..")]

from helpers import ARG_ORDER, edge_cases, make_batch, oracle_verify, to_wire, torsion_grid  # noqa: E402


def main() -> None:
    import torch
    import jubjub_schnorr_amd as jjs
    from jubjub_schnorr_amd import _ffi
    _ffi.select_library(_ffi.PROFILING_LIB_PATH)
    eng = jjs.engine()
    lib = _ffi.lib()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for scheme in ("single", "double", "vargen"):
        cases = [make_batch(scheme, n, seed=700 + n, n_keys=16) for n in (1, 3, 63, 65, 257, 1000, 5000)]
        cases += [edge_cases(scheme), torsion_grid(scheme, reps=2, extra=0 if scheme == "single" else 100)]
        for b in cases:
            want = oracle_verify(scheme, b)
            for path in (1, 0x42, 0x82, 0xF2):        # throughput path; latency path with 4, 8 and 16 pieces
                assert lib.jjs_debug_force_path(path) == 0
                st, tally = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]])
                assert st.cpu().numpy().tolist() == want.tolist(), (scheme, len(want), path)
                assert tally.cpu().numpy().tolist() == [int((want == k).sum()) for k in range(4)], (scheme, len(want), path)
                _, t2 = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]], want_status=False)
                assert t2.cpu().numpy().tolist() == tally.cpu().numpy().tolist()
                st_h, tally_h = eng.verify(scheme, *[b[k] for k in ARG_ORDER[scheme]])           # host buffers
                assert st_h.tolist() == want.tolist() and tally_h.tolist() == tally.cpu().numpy().tolist()
    # key-table path: both window widths on the same batch (218 signatures per key: the product takes the wide windows)
    kt_cases = {}
    for scheme in ("single", "double", "vargen"):
        b = make_batch(scheme, 65536 + 37, seed=808, n_keys=300)
        want = oracle_verify(scheme, b)
        kt_cases[scheme] = (b, want)
        for path in (0, 0x500):
            assert lib.jjs_debug_force_path(path) == 0
            st, tally = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]])
            assert (st.cpu().numpy() == want).all(), (scheme, path)
            assert tally.cpu().numpy().tolist() == [int((want == k).sum()) for k in range(4)], (scheme, path)
    lib.jjs_debug_force_path(0)
    # a device that cannot hold the key-table pool: the same batches take the throughput path, statuses unchanged, and the
    # path statistics say so (resident and host-buffer entry points)
    for scheme in ("single", "double", "vargen"):
        b, want = kt_cases[scheme]
        before = eng.path_stats()
        assert lib.jjs_debug_fail_key_arena(1) == 0
        st, tally = eng.verify(scheme, *[dev(b[k]) for k in ARG_ORDER[scheme]])
        st_h, tally_h = eng.verify(scheme, *[b[k] for k in ARG_ORDER[scheme]])
        assert lib.jjs_debug_fail_key_arena(0) == 0
        assert (st.cpu().numpy() == want).all() and (st_h == want).all(), scheme
        assert tally.cpu().numpy().tolist() == tally_h.tolist() == [int((want == k).sum()) for k in range(4)], scheme
        after = eng.path_stats()
        assert after["throughput"] == before["throughput"] + 2, (before, after)
        assert after["key_tables_wide"] + after["key_tables_narrow"] == before["key_tables_wide"] + before["key_tables_narrow"]
        # the wire form from host buffers, three times over (double signatures: the key columns travel behind the signatures):
        # without the key kernels every range decodes its own keys, once they have arrived
        w3 = [np.concatenate([a, a, a]) for a in to_wire(scheme, b)]
        assert lib.jjs_debug_fail_key_arena(1) == 0
        st_w, tally_w = eng.verify_wire(scheme, *w3)
        assert lib.jjs_debug_fail_key_arena(0) == 0
        assert (st_w == np.concatenate([want, want, want])).all(), scheme
        assert eng.path_stats()["throughput"] == after["throughput"] + 1
    # keys crafted to collide in the dedup table: harmless under the per-call seed of the product; with the seed pinned
    # (what the sender would need to know) the probe limit sends the batch down the throughput path, statuses unchanged
    from helpers import crafted_collision_batch
    b = crafted_collision_batch(n_good=1 << 16)
    want = oracle_verify("single", b)
    for pinned in (0, 1):
        before = eng.path_stats()
        assert lib.jjs_debug_pin_hash_seed(pinned) == 0
        st, tally = eng.verify("single", *[dev(b[k]) for k in ARG_ORDER["single"]])
        assert (st.cpu().numpy() == want).all(), pinned
        after = eng.path_stats()
        assert after["keys_probe_limit"] - before["keys_probe_limit"] == pinned, (pinned, before, after)
    lib.jjs_debug_pin_hash_seed(0)
    print("FORCEPATH OK")


if __name__ == "__main__":
    main()
