#!/bin/bash
# What the SGPR spills of prepare_kernel can cost: VALU wave-instructions per 64 hashes of
# challenge_kernel (hash alone, no spilled SGPRs) against prepare_kernel's head launch (hash + encodings + the cheap
# checks of R, 180 spilled SGPRs) on the same 2^20-item batch.  Usage (through gpurun): bash scripts/sgpr_spill_cost.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r03}; D=$R/gpurun_out/spill_$T; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $D -- python3 $R/jubjub_schnorr_amd/tools/hash_only.py > $D.log 2>&1 || exit 1
cd $R && python3 - "$D" <<'PY' | tee gpurun_out/sgpr_spill_cost_$(basename $D).json
import csv, glob, json, sys
c = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("challenge_kernel", "prepare_kernel"):
            if "::" + k + "(" in r["Kernel_Name"] and float(r["Counter_Value"]) > 1e6:
                c.setdefault(k, []).append(float(r["Counter_Value"]) / (2**20 / 64))
out = {k: round(max(v)) for k, v in c.items()}        # the largest launch of each: challenge of 2^20 items, prepare head of 2^20
out["difference"] = out.get("prepare_kernel", 0) - out.get("challenge_kernel", 0)
out["unit"] = "VALU wave-instructions per 64 single signatures"
out["note"] = ("prepare_kernel (head launch) = the same two permutations + encodings of 6 elements + curve / identity checks of R; "
               "the difference bounds everything that is not the hash, spill traffic included")
print(json.dumps(out))
PY
