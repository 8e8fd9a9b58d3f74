#!/bin/bash
# Kernel times of the shadow pass with ARCTIC_OPT_SMALL_TRIANGLES off and on (one rocprofv3 kernel trace: the first 40 k_raster<true> / k_setup launches are "off",
# the rest "on"), then the timing A/B, for each library given (default: the in-tree one).  Run on the GPU box from the repo root:
#     gpurun -- 'bash tools/experiments/small_ab.sh [build_tmp/lib_x.so ...] > gpurun_out/small_ab.txt'
set -e -o pipefail
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "${@:-arctic-renderer_amd/csrc/libarctic_hip.so}"; do
    export ARCTIC_HIP_LIBRARY="$ROOT/$lib"
    out=/tmp/small_ab_$$_$(basename "$lib")
    TRACE=1 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/tools/experiments/small_ab.py" > /dev/null 2>&1
    f=$(find "$out" -name "*kernel_trace.csv" | head -1)
    python3 - "$f" "$lib" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    for k in ("k_raster<true", "k_setup<true", "k_setup<false", "k_setup_clipped", "k_vertex", "k_shadow"):
        if k in n: per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2])
for k, v in per.items():
    # the trace holds, per handle, 10 warm-up + 30 traced passes: "off" first (k_setup<false>), then "on" (k_setup<true>)
    if k.startswith("k_setup<"): print(f"    {k:18s} launches {len(v):4d}  median of the last 30: {sorted(v[-30:])[15]:8.2f} us")
    else:
        h = len(v) // 2
        print(f"    {k:18s} launches {len(v):4d}  off: median of the last 30 {sorted(v[h - 30:h])[15]:8.2f} us   on: {sorted(v[-30:])[15]:8.2f} us")
PY
    python3 "$ROOT/tools/experiments/small_ab.py" 3
done
