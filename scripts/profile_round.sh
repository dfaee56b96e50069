#!/bin/bash
# One profiling pass on the GPU box: kernel trace, PMC passes (separate runs per counter group, as
# MI355X_MICROARCH.md prescribes), benches, phase profile with the gather ablation, clock / power sample.
# Usage (through gpurun): bash scripts/profile_round.sh <tag> [schemes]  -> files under gpurun_out/, summaries in
# profiles/<tag>_pmc_summary*.json and profiles/pmc_latest.json (stamped with the csrc hash)
# PART=1: the counter passes and benches only; PART=2: the small-call records, the timeline and the clock sample only (a pass
# is longer than one gpurun call allows; part 2 needs part 1's files under gpurun_out/ only for the last two lines of output)
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r04}; SCHEMES=${2:-"single double vargen"}; cd /tmp && export TMPDIR=/tmp
if [ "${PART:-0}" != 2 ]; then
for S in $SCHEMES single_unique; do
  KEYS=""; U=0; SS=$S
  if [ "$S" = "single_unique" ]; then SS=single; KEYS="--keys 1048576"; U=1; fi
  B="python3 $R/bench.py --scheme $SS $KEYS --no-cpu-baseline --no-two-streams --no-host-buffers"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_${S}_trace -- $B --steps 5 --warmup 1 > $R/gpurun_out/prof_${T}_${S}_trace.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_${S}_fetch -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${T}_${S}_fetch.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_${S}_write -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${T}_${S}_write.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/prof_${T}_${S}_sq -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${T}_${S}_sq.log 2>&1 &&
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 --output-format csv -d $R/gpurun_out/prof_${T}_${S}_int -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${T}_${S}_int.log 2>&1 || exit 1
  (cd $R && JJS_PMC_SCHEME=$SS JJS_PMC_UNIQUE_KEYS=$U python jubjub_schnorr_amd/tools/pmc_summary.py $T "$T" gpurun_out/prof_${T}_${S}_trace gpurun_out/prof_${T}_${S}_fetch gpurun_out/prof_${T}_${S}_write gpurun_out/prof_${T}_${S}_sq gpurun_out/prof_${T}_${S}_int > gpurun_out/pmc_${T}_${S}.log 2>&1
   find gpurun_out/prof_${T}_${S}_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_kernel_stats_${S}.csv \; )
done
cd $R
cp profiles/${T}_pmc_summary*.json profiles/pmc_latest.json gpurun_out/
# the bench line the driver will record (all three schemes, CPU baselines, multisig, small host calls), then wire / ext inputs
timeout -k 10 500 python bench.py > gpurun_out/bench_${T}.json 2> gpurun_out/bench_${T}.err
cp gpurun_out/bench_full_n1.json gpurun_out/bench_${T}_full.json
timeout -k 10 200 python bench.py --wire --no-cpu-baseline > gpurun_out/bench_${T}_wire.json 2>/dev/null
timeout -k 10 200 python bench.py --ext --no-cpu-baseline > gpurun_out/bench_${T}_ext.json 2>/dev/null
timeout -k 10 200 python bench.py --scheme multisig > gpurun_out/bench_${T}_multisig.json 2>/dev/null
for s in single double vargen; do timeout -k 10 200 python jubjub_schnorr_amd/tools/phase_profile.py $s 20 2>/dev/null >> gpurun_out/phase_profile_${T}.jsonl; done
for s in single double vargen; do timeout -k 10 200 python jubjub_schnorr_amd/tools/batch_size_curve.py $s > gpurun_out/batch_size_curve_${T}_$s.jsonl 2>/dev/null; done
timeout -k 10 200 python jubjub_schnorr_amd/tools/concurrent_calls.py > gpurun_out/concurrent_calls_${T}.jsonl 2>/dev/null
for s in single double vargen; do timeout -k 10 200 python -m jubjub_schnorr_amd.tools.tail_bound $s 3 >> gpurun_out/tail_bound_${T}.jsonl 2>/dev/null; done
fi   # PART != 2
cd $R
if [ "${PART:-0}" != 1 ]; then
bash scripts/r04_small_calls.sh ${T} full
bash scripts/timeline.sh ${T} single
bash scripts/clock_sample.sh $T
fi
cut -c1-200 gpurun_out/bench_${T}.json; tail -3 gpurun_out/pmc_${T}_single.log
