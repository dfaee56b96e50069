#!/bin/bash
# Copy what scripts/profile_round.sh <tag> left in gpurun_out/ into profiles/ (the tracked, judged copies).
T=${1:?tag}
cd "$(dirname "$0")/.."
python jubjub_schnorr_amd/tools/pmc_summary.py $T "${2:-$T}" gpurun_out/prof_${T}_trace gpurun_out/prof_${T}_fetch gpurun_out/prof_${T}_write gpurun_out/prof_${T}_sq > /dev/null
cp gpurun_out/${T}_kernel_stats.csv profiles/${T}_kernel_stats.csv
for s in single double vargen; do cp gpurun_out/bench_${T}_$s.json profiles/${T}_bench_$s.json; cp gpurun_out/bench_${T}_${s}_wire.json profiles/${T}_bench_${s}_wire.json; done
cp gpurun_out/phase_profile_${T}.jsonl profiles/${T}_phase_profile.jsonl
cp gpurun_out/host_rate_${T}.json profiles/${T}_host_buffer_rate.json
cp gpurun_out/multisig_rate_${T}.jsonl profiles/${T}_multisig_rate.jsonl
ls profiles | grep "^${T}_"
