#!/bin/bash
# Same-box A/B of the thread scaling of small host calls: the product build and every variant build libjjs_gpu_<name>.so beside
# it, alternating, $2 rounds.  $1: tag; $3: items per call; $4: thread counts.
set -o pipefail
mkdir -p gpurun_out
T=${1:-threads_ab}; R=${2:-2}; N=${3:-1}; TH=${4:-1,8,16,32,64}
for rep in $(seq $R); do
  for lib in jubjub_schnorr_amd/libjjs_gpu.so jubjub_schnorr_amd/libjjs_gpu_*.so; do
    case $lib in *_prof.so|*_trace*.so) continue;; esac
    timeout -k 10 200 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single --formats affine --sizes 1 --threads $TH --items-per-call $N --calls 200 --c-client --lib $lib 2>> gpurun_out/$T.err | grep threads >> gpurun_out/$T.jsonl || exit 1
  done
done
python - <<PY
import json
for l in open("gpurun_out/$T.jsonl"):
    r=json.loads(l); print(r["lib"], r["items_per_call"], {k: round(v) for k, v in r["calls_per_s"].items()}, r["calls_per_launch"])
PY
