#!/bin/bash
# Round 4: the blocking host-buffer entry points at small sizes and from several threads: the product build and whatever
# variant builds (libjjs_gpu_<name>.so) lie beside it.  $1: tag of the output files; $2: "full" = every scheme and format.
# Thread scaling comes from the C client (pthreads; tests/c/thread_client.c): python threads queue up at the interpreter lock.
set -o pipefail
mkdir -p gpurun_out
T=${1:-r04}
if [ "$2" = full ]; then
  timeout -k 10 500 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single,double,vargen --threads 1 > gpurun_out/${T}_small_host_calls.jsonl 2> gpurun_out/${T}_small_host_calls.err || exit 1
  timeout -k 10 500 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single,double,vargen --c-client >> gpurun_out/${T}_small_host_calls.jsonl 2>> gpurun_out/${T}_small_host_calls.err || exit 1
  # the reference's own API from a busy service: one signature (and 64) per call, up to 64 threads
  for n in 1 64; do
    timeout -k 10 300 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single --formats affine --sizes 1 --threads 1,2,4,8,16,32,64 --items-per-call $n --c-client 2>> gpurun_out/${T}_small_host_calls.err | grep threads >> gpurun_out/${T}_small_host_calls.jsonl || exit 1
  done
else
  timeout -k 10 300 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single --formats affine --sizes 1,1024,4096 --c-client > gpurun_out/${T}_small_host_calls.jsonl 2> gpurun_out/${T}_small_host_calls.err || exit 1
fi
for lib in jubjub_schnorr_amd/libjjs_gpu_*.so; do
  v=$(basename $lib .so); v=${v#libjjs_gpu_}
  [ "$v" = prof ] && continue
  timeout -k 10 300 python -m jubjub_schnorr_amd.tools.small_host_calls --schemes single --formats affine --sizes 1024 --c-client --lib $lib \
      > gpurun_out/${T}_small_host_calls_$v.jsonl 2> gpurun_out/${T}_small_host_calls_$v.err || exit 1
done
echo done
