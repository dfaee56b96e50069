#!/bin/bash
# One line about the box this runs on: median shader clock and board power while 2 500 back-to-back 2^20-item single
# batches run, and their rate.  Appended over several gpurun calls to profiles/<tag>_box_spread.jsonl (every call lands
# on another box of the pool).  Usage (through gpurun): bash scripts/box_record.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r02}; cd $R
bash scripts/clock_sample.sh box_$T > /dev/null 2>&1
python3 - "$T" <<'PY'
import json, sys
T = sys.argv[1]
rows = [json.loads(l) for l in open(f"gpurun_out/clock_power_box_{T}.jsonl") if l.strip()]
sc, pw = [], []
for r in rows:
    c = (r.get("rocm_smi") or {}).get("card0", {})
    try:
        s, p = int(c["sclk clock speed:"].strip("()Mhz")), float(c["Current Socket Graphics Package Power (W)"])
    except (KeyError, ValueError):
        continue
    if p > 800:                     # samples taken while the batches run
        sc.append(s); pw.append(p)
b = json.load(open(f"gpurun_out/clock_power_box_{T}_bench.json"))
med = lambda v: sorted(v)[len(v) // 2] if v else None
print(json.dumps({"sclk_mhz_median": med(sc), "power_w_median": med(pw), "samples": len(sc), "verifications_per_s": b["value"],
                  "ms_per_step": b["ms_per_step"], "steps": b["steps"], "bit_exact": b["bit_exact"]["status_vs_construction"]}))
PY
