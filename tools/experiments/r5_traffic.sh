#!/bin/bash
# Where the fabric-side read traffic of the shading pass comes from: FETCH_SIZE and the L2's hit / miss / request counters
# of k_material over tools/prof_shade.py full, for the options given.   usage: bash tools/experiments/r5_traffic.sh <tag> [lib.so] [name=value ...]
TAG=${1:?tag}; shift
ROOT=$(pwd)
if [ -n "$1" ] && [ "${1%.so}" != "$1" ]; then export ARCTIC_HIP_LIBRARY=$(realpath "$1"); export ARCTIC_HIP_LIBRARY_OLDER=1; shift; fi
OUT=$ROOT/gpurun_out/traffic_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/prof_shade.py" full 3 "$@" > "$OUT/pmc$i.log" 2>&1 || echo "[traffic] pass $i ($set) failed: $(tail -2 $OUT/pmc$i.log)"
    echo "[traffic] $TAG pass $i done: $set"
done
cd "$ROOT"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
root, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_material" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
out = {c: sum(v.values()) / len(v) for c, v in sorted(acc.items())}
print(json.dumps({"tag": tag, "per_launch": out}, indent=1))
json.dump({"tag": tag, "per_launch": out}, open(os.path.join(root, "summary.json"), "w"), indent=1)
PY
