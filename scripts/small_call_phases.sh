#!/bin/bash
# One small call under the kernel tracer: the durations of small_a_kernel (hash | chains | point checks side by side) and
# small_b_kernel, with the whole call and with the challenge hash skipped (profiling build) -- which of phase A's roles the
# call waits for.  Usage (through gpurun): bash scripts/small_call_phases.sh <tag> [n] [schemes] [variant lib] [skips]
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r04}; N=${2:-64}; SCHEMES=${3:-single double vargen}; LIB=${4:-}; SKIPS=${5:-0 2}; [ -n "$LIB" ] && LIB=$R/$LIB; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${T}_small_call_phases.jsonl
for s in $SCHEMES; do for skip in $SKIPS; do
  D=$R/gpurun_out/small_phases_${T}_${s}_$skip; rm -rf $D
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/scripts/small_call_trace.py $s $N $skip $LIB > $D.log 2>&1 || exit 1
  python3 - "$(find $D -name '*kernel_trace.csv' | head -1)" $s $N $skip "$LIB" >> $OUT <<'PY'
import csv, json, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = {}
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if n.startswith("small_"):
        d.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(json.dumps({"scheme": sys.argv[2], "items": int(sys.argv[3]), "hash_skipped": sys.argv[4] == "2", "lib": sys.argv[5].split("/")[-1] or "product",
                  "median_us": {k: round(statistics.median(v[5:]), 1) for k, v in d.items()}, "launches": {k: len(v) for k, v in d.items()}}))
PY
done; done
cat $OUT
