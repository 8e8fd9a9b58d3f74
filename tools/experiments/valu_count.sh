#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_VALU_TRANS_F32 of k_material per launch for a list of library builds (config 3, tools/prof_shade.py full):
# what an edit did to the instruction count, without the noise of a timing.   usage: bash tools/experiments/valu_count.sh lib1.so lib2.so ...
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/valu_count; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    name=$(basename "$lib" .so)
    ARCTIC_HIP_LIBRARY=$(realpath "$ROOT/$lib") ARCTIC_HIP_LIBRARY_OLDER=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/prof_shade.py" full 3 tile_order=0 > "$OUT/$name.log" 2>&1
    python3 - "$OUT/$name" "$name" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_material" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(sys.argv[2], {c: round(sum(v.values()) / len(v)) for c, v in sorted(acc.items())})
PY
done
