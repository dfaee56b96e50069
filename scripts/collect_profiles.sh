#!/bin/bash
# Copies the summaries of one profiling pass (scripts/profile_round.sh <tag>, merged back under gpurun_out/) into
# profiles/.  Usage: bash scripts/collect_profiles.sh <tag>
T=${1:?tag}; cd "$(dirname "$0")/.."
cp gpurun_out/pmc_latest.json gpurun_out/${T}_pmc_summary*.json profiles/
cp gpurun_out/bench_$T.json profiles/${T}_bench.json
cp gpurun_out/bench_${T}_full.json profiles/${T}_bench_full.json
cp gpurun_out/bench_${T}_wire.json profiles/${T}_bench_wire.json
cp gpurun_out/bench_${T}_ext.json profiles/${T}_bench_ext.json
cp gpurun_out/bench_${T}_multisig.json profiles/${T}_bench_multisig.json
for s in single double vargen single_unique; do cp gpurun_out/${T}_kernel_stats_$s.csv profiles/; done
cp gpurun_out/phase_profile_$T.jsonl profiles/${T}_phase_profile.jsonl
for s in single double vargen; do cp gpurun_out/batch_size_curve_${T}_$s.jsonl profiles/${T}_batch_size_curve_$s.jsonl; done
cp gpurun_out/clock_power_$T.jsonl profiles/${T}_clock_power.jsonl
cp gpurun_out/clock_power_${T}_bench.json profiles/${T}_clock_power_bench.json
cp gpurun_out/concurrent_calls_$T.jsonl profiles/${T}_concurrent_calls.jsonl
cp gpurun_out/tail_bound_$T.jsonl profiles/${T}_tail_bound.jsonl
cp gpurun_out/${T}_small_host_calls.jsonl profiles/${T}_small_host_calls.jsonl
cp gpurun_out/timeline_${T}_single.txt profiles/${T}_timeline_single.txt
python3 - "$T" <<'PY'
import json, sys
sys.path.insert(0, "."); import bench
T = sys.argv[1]
print("pmc hash matches csrc:", json.load(open("profiles/pmc_latest.json"))["csrc_sha256"] == bench.csrc_hash())
d = json.load(open(f"profiles/{T}_bench.json"))
r = d["roofline"]
print("single", d["value"], d["ms_per_step"], "traffic", r["traffic"], "binding", r.get("binding_frac"), (r.get("binding") or {}).get("per_kernel"))
print("host buffers", r.get("host_buffer"), "two streams", d.get("pipelined_two_streams"))
for k, v in d["schemes"].items():
    print(k, v["value"], v["ms_per_step"], v.get("traffic"), v.get("binding_frac"), v.get("host_buffer"), v.get("cpu_baseline"))
for k in ("unique_keys", "single_2p21", "single_all_valid", "multisig"):
    print(k, d.get(k))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["one_thread"]["value"])
print("small host calls", json.dumps(d.get("small_host_calls"))[:600])
for f in (f"{T}_bench_wire.json", f"{T}_bench_ext.json"):
    w = json.load(open("profiles/" + f)); print(f, w["value"], {k: v["value"] for k, v in w.get("schemes", {}).items()})
rows = [json.loads(l) for l in open(f"profiles/{T}_clock_power.jsonl")]
sc = sorted(int(r["rocm_smi"]["card0"]["sclk clock speed:"].strip("()Mhz")) for r in rows); pw = sorted(float(r["rocm_smi"]["card0"]["Current Socket Graphics Package Power (W)"]) for r in rows)
print("sclk", sc[2], sc[len(sc) // 2], sc[-3], "power", pw[3], pw[len(pw) // 2], pw[-1])
b = json.load(open(f"profiles/{T}_clock_power_bench.json")); print("long run", b["value"], b["ms_per_step"], b["steps"])
PY
