#!/bin/bash
# A/B timing on ONE box (boxes differ by ~2 % in clock): bench the in-tree library and a variant build
# alternately.  Usage (through gpurun): bash scripts/ab_bench.sh /path/to/variant.so [scheme] [rounds] [further bench flags, e.g. --ext]
B=${1:?variant .so}; S=${2:-single}; N=${3:-3}; X=${4:-}
for i in $(seq $N); do
  a=$(python bench.py --no-cpu-baseline --no-host-buffers --no-two-streams --scheme $S $X 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python bench.py --lib $B --no-cpu-baseline --no-host-buffers --no-two-streams --scheme $S $X 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "round $i: A(in-tree) $a ms   B(variant) $b ms"
done
