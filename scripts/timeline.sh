#!/bin/bash
# Kernel timeline of one 2^20-item batch (start / end of every kernel relative to the batch's first, with its stream):
# what runs beside what.  Usage (through gpurun): bash scripts/timeline.sh <tag> [scheme] [extra bench flags] [label]
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r02}; SCHEME=${2:-single}; X=${3:-}; S=${4:-$SCHEME}; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/timeline_${T}_$S -- python3 $R/bench.py --scheme $SCHEME $X --no-cpu-baseline --no-two-streams --no-host-buffers --steps 3 --warmup 2 > $R/gpurun_out/timeline_${T}_$S.log 2>&1 || exit 1
python3 - "$(find $R/gpurun_out/timeline_${T}_$S -name '*kernel_trace.csv' | head -1)" > $R/gpurun_out/timeline_${T}_$S.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
# the last batch: from the last key_dedup / prepare launch that follows a resolve_kernel
last_resolve = [i for i, r in enumerate(rows) if names(r) == "resolve_kernel"]
start = last_resolve[-2] + 1 if len(last_resolve) >= 2 else 0
batch = [r for r in rows[start:last_resolve[-1] + 1] if not names(r).startswith(("void at::", "__amd_rocclr"))]
t0 = min(int(r["Start_Timestamp"]) for r in batch)
print("kernel                    stream  queue   start_ms   end_ms   dur_ms")
for r in batch:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{names(r):25s} {r['Stream_Id']:>6s} {r['Queue_Id']:>6s} {s:10.3f} {e:8.3f} {e - s:8.3f}")
PY
cat $R/gpurun_out/timeline_${T}_$S.txt
