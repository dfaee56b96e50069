#!/usr/bin/env python3
"""Per-phase time of the verify kernel by ablation (skip one phase, time the rest) on the bench batch.
Usage: python jubjub_schnorr_amd/tools/phase_profile.py [scheme] [log2n]  -> one JSON line."""
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
    scheme = sys.argv[1] if len(sys.argv) > 1 else "single"
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    _ffi.select_library(_ffi.PROFILING_LIB_PATH)      # the ablation switches exist only in the -DJJS_PROFILING build
    eng = jjs.engine()
    _ffi.check(_ffi.lib().jjs_debug_force_path(3), "force_path")      # the throughput path, without key tables, is what is profiled
    arrays, _ = bench.make_inputs(eng, scheme, 1 << log2n, 0)
    call = [arrays[k] for k in bench.ARG_ORDER[scheme]]

    def timed(mask):
        _ffi.check(_ffi.lib().jjs_debug_skip_phases(mask), "skip")
        eng.verify(scheme, *call)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.verify(scheme, *call); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    # bit0: point checks (cheap checks, pairing tests, hence no resolve pass); bit1: challenge; bit2: equations;
    # bit3: Euclid.  Differences are taken between runs that agree on everything else, and never against a run
    # whose equations all fail (that would send every item through the resolve pass).
    full = timed(0)
    out = {"scheme": scheme, "items": 1 << log2n, "ms_full": full,
           "ms_without_validity": timed(1), "ms_without_equations": timed(4),
           "ms_without_equations_and_challenge": timed(6), "ms_without_validity_and_equations": timed(5),
           "ms_without_validity_and_euclid": timed(9), "ms_only_loads_and_tally": timed(7)}
    out["ms_validity"] = full - out["ms_without_validity"]
    out["ms_challenge"] = out["ms_without_equations"] - out["ms_without_equations_and_challenge"]
    out["ms_equations"] = out["ms_without_validity"] - out["ms_without_validity_and_equations"]
    out["ms_euclid"] = out["ms_without_validity"] - out["ms_without_validity_and_euclid"]
    # the gathers: window tables in a 340 MB workspace and comb rows in a 117 MB table, against lookups whose working
    # set stays in L2 (bit 4: entry 1 of the lane's own table, 256 entries per comb row; point checks off in both runs
    # so that no item goes through the resolve pass)
    out["ms_without_validity_lookups_cache_resident"] = timed(1 | 16)
    out["ms_gathers"] = out["ms_without_validity"] - out["ms_without_validity_lookups_cache_resident"]
    # the key-table path: the product groups the items by key, so that a wave looks up one or two keys' tables;
    # 0x1000 keeps the caller's order (64 keys per wave, ~1 GB of tables touched at random); bit 4 there confines the
    # comb lookups to 256 entries per row (point checks off in both runs: no resolve pass)
    _ffi.check(_ffi.lib().jjs_debug_force_path(1), "force_path")
    out["ms_key_table_path"] = timed(0)
    out["ms_key_table_path_no_validity"] = timed(1)
    out["ms_key_table_path_no_validity_comb_cache_resident"] = timed(1 | 16)
    out["ms_key_table_comb_gathers"] = out["ms_key_table_path_no_validity"] - out["ms_key_table_path_no_validity_comb_cache_resident"]
    _ffi.check(_ffi.lib().jjs_debug_force_path(1 | 0x1000), "force_path")
    out["ms_key_table_path_callers_order"] = timed(0)
    out["ms_key_table_grouping_gain"] = out["ms_key_table_path_callers_order"] - out["ms_key_table_path"]
    _ffi.lib().jjs_debug_force_path(0)
    _ffi.lib().jjs_debug_skip_phases(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
