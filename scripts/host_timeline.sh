#!/bin/bash
# Timeline of one blocking host-buffer call of 2^20 items (kernels and copies of the last of three calls, relative to its
# first upload): where the 2-3 ms between a host-buffer call and a resident batch go.
# Usage (through gpurun): bash scripts/host_timeline.sh <tag> [scheme] [format: affine|ext|wire]
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r03}; SCHEME=${2:-single}; F=${3:-affine}; cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/host_timeline_${T}_${SCHEME}_$F
export JJS_HOST_CALLS_ONLY=1
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -- python3 $R/jubjub_schnorr_amd/tools/host_calls.py $SCHEME $F 3 > $D.log 2>&1 || exit 1
python3 - "$(find $D -name '*kernel_trace.csv' | head -1)" "$(find $D -name '*memory_copy_trace.csv' | head -1)" > $D.txt <<'PY'
import csv, sys
ks = list(csv.DictReader(open(sys.argv[1])))
cs = list(csv.DictReader(open(sys.argv[2])))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), "s" + r["Stream_Id"]) for r in ks
      if not name(r).startswith(("void at::", "__amd_rocclr"))]
for r in cs:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r["Direction"].replace("MEMORY_COPY_", "") + " %.1f MB" % (int(r.get("Size", 0) or 0) / 1e6), "s" + r.get("Stream_Id", "?")))
ev.sort()
# the last call: everything after the second-to-last resolve_kernel
res = [i for i, e in enumerate(ev) if e[2] == "resolve_kernel"]
start = res[-2] + 1 if len(res) >= 2 else 0
call = [e for e in ev[start:] if e[0] <= ev[res[-1]][1] + 2_000_000]
t0 = min(e[0] for e in call)
print("what                              stream   start_ms   end_ms   dur_ms")
for s, e, n, st in call:
    print(f"{n:34s} {st:>6s} {(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:8.3f}")
PY
cat $D.txt; tail -3 $D.log
