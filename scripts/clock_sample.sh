#!/bin/bash
# Shader clock, memory clock and board power sampled once a second while the single-scheme bench loop runs
# (about 8 s of back-to-back 2^20-item batches): the record that goes with the PMC summary's effective clock.
# Usage (through gpurun): bash scripts/clock_sample.sh <tag>   -> gpurun_out/clock_power_<tag>.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r02}; OUT=$R/gpurun_out/clock_power_${T}.jsonl; : > $OUT
cd $R
timeout -k 10 200 python bench.py --scheme single --no-cpu-baseline --steps 2500 --warmup 5 > gpurun_out/clock_power_${T}_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 150); do        # one sample a second for as long as the bench process lives (import, inputs, ~26 s of batches)
  if ! kill -0 $BP 2>/dev/null; then break; fi
  S=$(rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | tr -d '\n')
  if [ -n "$S" ]; then echo "{\"t\": $i, \"rocm_smi\": $S}" >> $OUT; fi
  sleep 1
done
wait $BP
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
for r in rows[:3] + rows[-1:]:
    c = (r.get("rocm_smi") or {}).get("card0", {})
    print({k: v for k, v in c.items() if "sclk" in k.lower() or "Power" in k or "mclk" in k.lower()})
PY
