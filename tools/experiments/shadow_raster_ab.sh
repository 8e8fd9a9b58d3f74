#!/bin/bash
# Kernel times of the shadow pass for several builds of the library, one rocprofv3 kernel trace each (run on the GPU box from the repo root):
#     gpurun -- 'bash tools/experiments/shadow_raster_ab.sh build_tmp/lib_a.so build_tmp/lib_b.so > gpurun_out/raster_ab.txt'
set -e -o pipefail
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
    export ARCTIC_HIP_LIBRARY="$ROOT/$lib"
    out=/tmp/raster_ab_$$_$(basename "$lib")_$rep
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$ROOT/tools/experiments/shadow_pass_loop.py" 2> /dev/null | grep "shadow pass"
    f=$(find "$out" -name "*kernel_stats.csv" | head -1)
    python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    for k in ("k_raster<true", "k_setup(", "k_setup_clipped", "k_vertex", "k_shadow_blocks", "k_shadow_bounds"):
        if k in n:
            print(f"    {k:18s} calls {row['Calls']:>4s}  avg {float(row['AverageNs']) / 1e3:8.2f} us  min {float(row['MinNs']) / 1e3:8.2f}")
PY
done
done
