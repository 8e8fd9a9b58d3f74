#!/bin/bash
# Counters of the shadow pass's kernels for one or more builds (separate --pmc passes, kernel trace only beside them):
#     gpurun -- 'bash tools/experiments/shadow_raster_pmc.sh build_tmp/lib_a.so ... > gpurun_out/raster_pmc.txt'
set -e -o pipefail
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    export ARCTIC_HIP_LIBRARY="$ROOT/$lib"
    echo "== $lib"
    i=0
    for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU" \
               "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
               "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
        i=$((i + 1))
        out=/tmp/raster_pmc_$$_$(basename "$lib")_$i
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -- python3 "$ROOT/tools/experiments/shadow_pass_loop.py" > /dev/null 2>&1 || { echo "pass $i failed: $set"; continue; }
        f=$(find "$out" -name "*counter_collection.csv" | head -1)
        python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_raster<true" not in k: continue
    acc["k_raster<true>"][row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
for k, d in acc.items():
    print("   ", k, "  ".join(f"{c} {v / max(1, len(n[c])):.4g}" for c, v in sorted(d.items())))
PY
    done
done
