#!/bin/bash
# One profiling pass on the GPU box: kernel trace, PMC passes, benches, phase profile.  Usage (through gpurun):
#   bash scripts/profile_round.sh <tag>      -> files under gpurun_out/, summary in profiles/<tag>_pmc_summary.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r01m}; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${T}_trace.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${T}_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${T}_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/prof_${T}_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${T}_sq.log 2>&1
cd $R
python jubjub_schnorr_amd/tools/pmc_summary.py $T "$T" gpurun_out/prof_${T}_trace gpurun_out/prof_${T}_fetch gpurun_out/prof_${T}_write gpurun_out/prof_${T}_sq > gpurun_out/pmc_${T}.log 2>&1
cp profiles/${T}_pmc_summary.json profiles/pmc_latest.json gpurun_out/
find gpurun_out/prof_${T}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_kernel_stats.csv \;
for s in single double vargen; do timeout -k 10 280 python bench.py --scheme $s > gpurun_out/bench_${T}_$s.json 2>/dev/null; timeout -k 10 280 python bench.py --scheme $s --wire --no-cpu-baseline > gpurun_out/bench_${T}_${s}_wire.json 2>/dev/null; done
for s in single double vargen; do timeout -k 10 200 python jubjub_schnorr_amd/tools/phase_profile.py $s 20 2>/dev/null >> gpurun_out/phase_profile_${T}.jsonl; done
timeout -k 10 200 python jubjub_schnorr_amd/tools/host_rate.py > gpurun_out/host_rate_${T}.json 2>/dev/null
timeout -k 10 300 python jubjub_schnorr_amd/tools/multisig_rate.py > gpurun_out/multisig_rate_${T}.jsonl 2>/dev/null
cut -c1-170 gpurun_out/bench_${T}_*.json; tail -12 gpurun_out/pmc_${T}.log
