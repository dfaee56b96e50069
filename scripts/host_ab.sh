#!/bin/bash
# A/B timing of host-buffer piece plans on ONE box: builds variants of the library with other JJS_HOST_* knobs
# (csrc/host_calls.h) HERE, before gpurun -- e.g.  bash scripts/host_ab.sh build v1 -DJJS_HOST_LEAD_SHARE_DEN=4 --
# and times them alternately there:  bash scripts/host_ab.sh run <scheme> <format> <rounds> v1 v2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
  name=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o $R/jubjub_schnorr_amd/libjjs_gpu_$name.so $R/jubjub_schnorr_amd/csrc/jjs_gpu.hip
  exit $?
fi
shift; S=$1; F=$2; N=$3; shift 3
for i in $(seq $N); do
  python3 $R/jubjub_schnorr_amd/tools/host_calls.py $S $F 7 20 2>/dev/null | tail -1
  for v in "$@"; do
    python3 $R/jubjub_schnorr_amd/tools/host_calls.py $S $F 7 20 $R/jubjub_schnorr_amd/libjjs_gpu_$v.so 2>/dev/null | tail -1
  done
done
