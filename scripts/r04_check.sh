#!/bin/bash
# Round 4 check on the GPU box: what the stretch of a batch that is not arithmetic is worth (tools/tail_bound.py), one bench
# line, then the whole -m gpu suite with its slowest tests.  $1: tag of the output files.
T=${1:-r04}
mkdir -p gpurun_out
timeout -k 10 200 python -m jubjub_schnorr_amd.tools.tail_bound single 5 > gpurun_out/${T}_tail_bound.jsonl 2>gpurun_out/${T}_tail_bound.err; cut -c1-700 gpurun_out/${T}_tail_bound.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-buffers > gpurun_out/${T}_bench_short.json 2>gpurun_out/${T}_bench_short.err; cut -c1-300 gpurun_out/${T}_bench_short.json
if [ "$2" != nosuite ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/${T}_gpu_suite.log 2>&1; echo suite rc=$?; tail -22 gpurun_out/${T}_gpu_suite.log
fi
