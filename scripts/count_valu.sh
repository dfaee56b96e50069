#!/bin/bash
# VALU wave-instructions per 64 verifications (prepare + verify + resolve kernels), from one PMC pass: the
# deterministic way to compare kernel variants (time = count x ~4.4 cycles; boxes differ by ~2 % in clock).
# Usage (through gpurun): bash scripts/count_valu.sh [scheme] [path/to/variant.so]
S=${1:-single}; LIB=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}; D=$R/gpurun_out/valu_$$; cd /tmp && export TMPDIR=/tmp
if [ -n "$LIB" ]; then LIBARG="--lib $LIB"; fi
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $D -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --scheme $S $LIBARG > /dev/null 2>&1
cd $R && python3 - "$D" <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("prepare_kernel", "verify_kernel", "resolve_kernel"):
            if k + "(" in r["Kernel_Name"]:
                tot.setdefault(k, []).append(float(r["Counter_Value"]))
s = sum(sum(v) / len(v) for v in tot.values())
print("VALU wave-instructions per 64 verifications: %.0f   (%s)" % (s / (2**20 / 64), {k: round(sum(v) / len(v) / 16384) for k, v in tot.items()}))
PY
rm -rf $D
