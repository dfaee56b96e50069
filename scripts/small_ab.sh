#!/bin/bash
# Same-box A/B of one small call: the product build and every variant build libjjs_gpu_<name>.so beside it, alternating,
# $2 rounds; the resident call's device time (dev_ms) and the blocking host call by size.  $1: tag of the output file.
set -o pipefail
mkdir -p gpurun_out
T=${1:-small_ab}; R=${2:-2}; SCHEMES=${3:-single,double,vargen}; SIZES=${4:-1,64,1024,4096}
for rep in $(seq $R); do
  for lib in jubjub_schnorr_amd/libjjs_gpu.so jubjub_schnorr_amd/libjjs_gpu_*.so; do
    case $lib in *_prof.so|*_trace*.so) continue;; esac
    timeout -k 10 200 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes $SCHEMES --formats affine --sizes $SIZES --threads 1 --calls 100 --c-client --lib $lib \
        >> gpurun_out/$T.jsonl 2>> gpurun_out/$T.err || exit 1
  done
done
python - <<PY
import json
rows=[json.loads(l) for l in open("gpurun_out/$T.jsonl") if l.startswith("{")]
for r in rows:
    if r["what"]=="latency":
        print(r["lib"], r["scheme"], "dev", r["ms_per_call"]["dev_ms"], "host", r["ms_per_call"]["affine"])
PY
