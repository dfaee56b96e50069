#!/bin/bash
# VGPRs / spills / occupancy of every kernel in csrc/jjs_gpu.hip, from the compiler (no GPU needed).
# Usage: bash scripts/kernel_resources.sh [extra hipcc flags, e.g. -DJJS_PROFILING]
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -o /dev/null "$@" \
    $R/jubjub_schnorr_amd/csrc/jjs_gpu.hip -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur, vals = None, {}
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        k = re.search(r"\d+([a-z_0-9]+_kernel)", name)
        cur, vals = (k.group(1) if k else name), {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        vals[m.group(1)] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            print("%-26s VGPR %3d  SGPR %3d (spilled %3d)  scratch %4d B/lane  VGPR spill %3d  waves/SIMD %d  LDS %d" % (
                cur, vals.get("VGPRs", -1), vals.get("TotalSGPRs", -1), vals.get("SGPRs Spill", -1), vals.get("ScratchSize", -1),
                vals.get("VGPRs Spill", -1), vals.get("Occupancy", -1), vals.get("LDS Size", -1)))
'
