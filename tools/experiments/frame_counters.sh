#!/bin/bash
# vector / scalar instruction counts per launch of EVERY kernel of a whole frame (tools/prof_shade.py render_frame, config 3): where a frame's issue slots go.
# On the GPU box, from the repo root:   bash tools/experiments/frame_counters.sh <out dir under gpurun_out> [option=value ...]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$1; shift; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/tools/prof_shade.py" render_frame 3 shadow_cache=0 "$@" > "$OUT/run.log" 2>&1
python3 - "$OUT/pmc" > "$OUT/counters.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("arctic::(anonymous namespace)::", "").split("(")[0]
        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_INSTS_VALU", {0: 0}).values())):
    print(f"{k:40s}", "  ".join(f"{c} {sum(v.values()) / len(v) / 1e6:8.3f} M" for c, v in sorted(cs.items())), f"  ({len(next(iter(cs.values())))} launches)")
PY
rm -rf "$OUT/pmc"
cat "$OUT/counters.txt"
