"""Assemble profiles/<tag>_pmc_summary.json and profiles/pmc_latest.json from rocprofv3 CSV output.

    python jubjub_schnorr_amd/tools/pmc_summary.py <tag> <variant text> <trace_dir> <pmc_dir> [<pmc_dir> ...]
    (JJS_PMC_SCHEME=double|vargen in the environment when the passes ran `bench.py --scheme <that>`)

<trace_dir>: output of  rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
<pmc_dir>s : outputs of rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py ...   (one pass per
             counter group, as MI355X_MICROARCH.md prescribes)
One batch = one launch of each kernel in KERNELS (throughput path: prepare, verify, resolve; key-table path: the
key_* kernels, prepare, key_verify, resolve, and a verify_kernel that leaves at once); counters are summed over them
and averaged over the batches of the run.  HBM bytes = FETCH_SIZE (KB) x 2 (gfx950 correction for 16 B/lane
loads) + WRITE_SIZE (KB).
"""
import csv
import glob
import json
import os
import sys

# every kernel one batch launches (the key-table path adds the key_* kernels; verify_kernel then leaves at once)
KERNELS = ("prepare_kernel", "verify_kernel", "resolve_kernel", "key_dedup_kernel", "key_assign_kernel", "key_spread_kernel",
           "key_count_kernel", "key_scan_kernel", "key_scatter_kernel", "key_chain_kernel", "key_table_kernel", "key_verify_kernel")
ONCE_PER_BATCH = "resolve_kernel"      # launched exactly once per batch: its launch count is the number of batches
DOMINANT = ("key_verify_kernel", "verify_kernel")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_of(name: str):
    for k in KERNELS:
        if "::" + k + "(" in name or name.startswith(k + "("):      # "verify_kernel(" is also a suffix of key_verify_kernel(
            return k
    return None


def read_counters(d):
    out = {}
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = kernel_of(row["Kernel_Name"])
            if not k:
                continue
            out.setdefault(row["Counter_Name"], {}).setdefault(k, []).append(float(row["Counter_Value"]))
            if k in DOMINANT and float(row["Counter_Value"]) > 0 and (k == "key_verify_kernel" or "equations_kernel" not in meta):
                meta = {"equations_kernel": k, "grid_size": int(row["Grid_Size"]), "lds_block_size": int(row["LDS_Block_Size"]),
                        "scratch_size": int(row["Scratch_Size"]), "vgpr_count": int(row["VGPR_Count"]),
                        "sgpr_count": int(row["SGPR_Count"])}
    return out, meta


def read_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = kernel_of(row["Name"])
            if k:
                out[k] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                          "min_ms": float(row["MinNs"]) / 1e6, "max_ms": float(row["MaxNs"]) / 1e6}
    return out


ALGO_BYTES = {"single": 196, "double": 324, "vargen": 260}


def main():
    tag, variant, trace_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
    scheme = os.environ.get("JJS_PMC_SCHEME", "single")
    unique = os.environ.get("JJS_PMC_UNIQUE_KEYS") == "1"       # the passes ran bench.py --keys <items>: throughput path
    sys.path.insert(0, ROOT)
    import bench
    stats = read_stats(trace_dir)
    counters, meta = {}, {}
    for d in pmc_dirs:
        c, m = read_counters(d)
        counters.update(c)
        meta = m or meta
    summary = {"round": 4, "variant": variant, "csrc_sha256": bench.csrc_hash(),
               "command": "rocprofv3 --pmc <C> --output-format csv -- python3 bench.py --scheme " + scheme + " --steps 3 --warmup 1 "
                          "--no-cpu-baseline (one pass per counter group); kernel times from a separate "
                          "rocprofv3 --kernel-trace --stats run",
               "kernels": list(KERNELS), "scheme": scheme, "items": 1 << 20, "counters": {}}
    per_batch = {}
    for name, by_kernel in sorted(counters.items()):
        entry = {}
        total = 0.0
        batches = len(by_kernel.get(ONCE_PER_BATCH, [])) or 1
        for k, vals in by_kernel.items():
            # a kernel may be launched more than once per batch (prepare_kernel: head and tail): sum over a batch
            entry[k] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals), "per_batch": sum(vals) / batches}
            total += sum(vals) / batches
        entry["per_batch"] = total
        per_batch[name] = total
        summary["counters"][name] = entry
    summary.update(meta)
    summary["kernel_ms_rocprof"] = stats
    # key_chain / key_table run on a second stream beside prepare_kernel: the sum of the kernel times is an upper
    # bound of the batch time (bench.py's HIP-event time is the batch time)
    n_batches = stats.get(ONCE_PER_BATCH, {}).get("calls", 0) or 1
    batch_ms = sum(v["avg_ms"] * v["calls"] / n_batches for v in stats.values())
    summary["batch_ms_rocprof"] = batch_ms
    summary["batch_ms_rocprof_note"] = "sum over kernels; key_chain_kernel and key_table_kernel overlap prepare_kernel"
    if "FETCH_SIZE" in per_batch and "WRITE_SIZE" in per_batch:
        summary["hbm_bytes_per_launch"] = (2.0 * per_batch["FETCH_SIZE"] + per_batch["WRITE_SIZE"]) * 1024.0
        summary["hbm_bytes_note"] = ("FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md HBM section (16 B/lane loads; "
                                     "gather-like access, so an upper estimate), WRITE_SIZE (KB) as is; verify + resolve")
    summary["algorithmic_bytes_per_launch"] = ALGO_BYTES[scheme] * (1 << 20)
    if "SQ_INSTS_VALU" in per_batch:
        valu = per_batch["SQ_INSTS_VALU"]
        summary["valu_wave_instr_per_launch"] = valu
        summary["valu_wave_instr_per_64_verifies"] = valu / ((1 << 20) / 64)
        if "SQ_INSTS_VALU_INT64" in per_batch:
            # the counter takes v_mad_u64_u32, v_mad_i64_i32 and the 64-bit shifts / adds (calibrated on the single-opcode
            # kernels of tools/microbench under the same counters: profiles/r03_pmc_counter_calibration.txt): the
            # instructions that occupy a SIMD for four cycles a wave
            summary["valu_int64_wave_instr_per_launch"] = per_batch["SQ_INSTS_VALU_INT64"]
            summary["valu_int64_share"] = per_batch["SQ_INSTS_VALU_INT64"] / valu
        if "SQ_INSTS_VALU_INT32" in per_batch:
            summary["valu_int32_wave_instr_per_launch"] = per_batch["SQ_INSTS_VALU_INT32"]
        overlapped = stats.get("key_verify_kernel", {}).get("avg_ms", 0.0) > 0.05      # key-table path: kernels on two streams
        if "GRBM_GUI_ACTIVE" in per_batch and batch_ms and not overlapped:
            cycles = per_batch["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
            summary["effective_clock_ghz"] = cycles / (batch_ms * 1e-3) / 1e9
            summary["cycles_per_valu_instr_per_simd"] = cycles * 1024 / valu
        elif overlapped:
            summary["effective_clock_ghz"] = None
            summary["effective_clock_note"] = ("not derived: key_chain / key_table overlap prepare_kernel, so neither the sum of "
                                               "kernel times nor the sum of per-dispatch busy cycles is the batch's; see the "
                                               "unique-keys summary (sequential kernels) and clock_power_*.jsonl")
    # Per kernel, from its OWN counters of the PMC passes (rocprofv3 serialises dispatches while it collects counters, so
    # these are the kernels one by one, not overlapped as in a batch): the issue bound 4 * I64 + 2 * (I - I64) SIMD-cycles
    # against the SIMD-cycles the dispatch had, 1 024 SIMDs x GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs) -- a
    # fraction that needs no clock.  `wait_inst_share`: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, the share of its resident time a
    # wave spent waiting to issue (with w waves on a SIMD that is bound by issue, (w - 1) / w of it).  `trace_ms`: the
    # dispatch in the batch as it runs (kernel trace, overlapped), `share_of_batch_time`: of the sum over kernels.
    per_kernel = {}
    total_trace = sum(v["avg_ms"] * v["calls"] / n_batches for v in stats.values()) or 1.0
    for k in KERNELS:
        valu = summary["counters"].get("SQ_INSTS_VALU", {}).get(k, {}).get("per_batch")
        if valu is None:
            continue
        i64 = summary["counters"].get("SQ_INSTS_VALU_INT64", {}).get(k, {}).get("per_batch", 0.0)
        grbm = summary["counters"].get("GRBM_GUI_ACTIVE", {}).get(k, {}).get("per_batch")
        wave = summary["counters"].get("SQ_WAVE_CYCLES", {}).get(k, {}).get("per_batch")
        wait = summary["counters"].get("SQ_WAIT_INST_ANY", {}).get(k, {}).get("per_batch")
        waves = summary["counters"].get("SQ_WAVES", {}).get(k, {}).get("per_batch")
        st = stats.get(k, {})
        trace_ms = st.get("avg_ms", 0.0) * st.get("calls", 0) / n_batches
        rec = {"valu_wave_instr": valu, "int64_share": (i64 / valu) if valu else None, "waves": waves, "trace_ms": trace_ms,
               "share_of_batch_time": trace_ms / total_trace}
        if grbm:
            cycles = grbm / 8.0
            rec["elapsed_cycles_alone"] = cycles
            rec["alu_frac_alone"] = (4.0 * i64 + 2.0 * (valu - i64)) / (1024.0 * cycles)
        if wave and wait is not None:
            rec["wait_inst_share"] = wait / wave
        per_kernel[k] = rec
    summary["per_kernel"] = per_kernel
    by_time = sorted(per_kernel, key=lambda k: -per_kernel[k]["trace_ms"])
    summary["dominant_kernel"] = by_time[0] if by_time else None
    summary["dominant_kernel_note"] = "by time in the batch (kernel trace); grid_size / vgpr_count / sgpr_count above are those of the equations kernel"
    if unique and "key_table_kernel" in per_kernel:
        summary["key_kernels_note"] = ("every signature under its own key: the key kernels decide against the tables and key_chain / key_table leave "
                                       "at once (their instruction counts above); their trace_ms is the span from the first to the last of their "
                                       "blocks being DISPATCHED on the lowest-priority streams, i.e. time spent waiting for wave slots beside "
                                       "prepare_kernel, not time spent running (elapsed_cycles_alone is what they take by themselves)")
    suffix = ("" if scheme == "single" else "_" + scheme) + ("_unique_keys" if unique else "")
    summary["keys"] = "every signature under its own key (throughput path)" if unique else "4096 key pairs (SURVEY.md 8d; key-table path)"
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary{suffix}.json")
    json.dump(summary, open(out, "w"), indent=1)
    # profiles/pmc_latest.json: what bench.py may quote -- only for the code it was measured on (csrc hash)
    latest_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        latest = json.load(open(latest_path))
    except (OSError, ValueError):
        latest = {}
    if latest.get("csrc_sha256") != summary["csrc_sha256"]:
        latest = {"csrc_sha256": summary["csrc_sha256"], "schemes": {}}
    latest["schemes"][scheme + ("_unique_keys" if unique else "")] = {"items": 1 << 20, "hbm_bytes_per_launch": summary.get("hbm_bytes_per_launch"),
                                 "valu_wave_instr_per_launch": summary.get("valu_wave_instr_per_launch"),
                                 "valu_int64_wave_instr_per_launch": summary.get("valu_int64_wave_instr_per_launch"),
                                 "dominant_kernel": summary.get("dominant_kernel"),
                                 "per_kernel": {k: {"alu_frac_alone": round(v["alu_frac_alone"], 4) if "alu_frac_alone" in v else None,
                                                    "share_of_batch_time": round(v["share_of_batch_time"], 4),
                                                    "valu_wave_instr": v["valu_wave_instr"]}
                                                for k, v in per_kernel.items() if v["share_of_batch_time"] >= 0.02},
                                 "source": f"profiles/{tag}_pmc_summary{suffix}.json"}
    json.dump(latest, open(latest_path, "w"), indent=1)
    print(json.dumps({k: summary[k] for k in summary if k not in ("counters",)}, indent=1))


if __name__ == "__main__":
    main()
