#!/bin/bash
# HBM traffic and duration of the shading pass of ANOTHER BASELINE configuration (profile_round.sh is the metric's config 3):
#     gpurun -- 'bash tools/profile_config.sh 2 r4_c2'
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (TCC has 4 counter slots), each with --kernel-trace only, over
# tools/prof_shade.py (5 launches of the pass; one handle -- the counters are per launch, the Infinity Cache does not hide HBM reads
# from FETCH_SIZE's point of view: it counts requests leaving L2).
set -e -o pipefail
CFG=${1:?usage: profile_config.sh <config> <tag>}
TAG=${2:?usage: profile_config.sh <config> <tag> [option=value ...]}
shift 2   # what is left: arctic_set_option pairs for tools/prof_shade.py
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --stats --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/prof_shade.py" full "$CFG" "$@" > "$OUT/pmc$i.log" 2>&1
    echo "[profile] config $CFG counters pass $i done: $set"
done
cd "$ROOT"
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root, cfg = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_material" in row["Kernel_Name"]:
            acc[row["Counter_Name"]][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
out = {c: sum(v.values()) / len(v) for c, v in sorted(acc.items())}
dur = {}
for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_material" in row["Name"]:
            dur[os.path.relpath(path, root).split(os.sep)[0]] = float(row["AverageNs"])
res = {"config": int(cfg), "counters": out, "kernel_ns_per_pass": dur}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    res["hbm_bytes_per_launch"] = int(round((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024))
    res["correction"] = "reads doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B); writes as reported"
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(root, "summary.json"), "w"), indent=1)
PY
