#!/bin/bash
# cache-side counters of the PCF slow-path A/B (register window against LDS tile), sun-only pass of config 3:
# rocprofv3 --pmc in separate runs per variant; output: gpurun_out/lds_ab/<variant>_<set>
ROOT=$(pwd); OUT=$ROOT/gpurun_out/lds_ab; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in dbg0 dbg16; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --stats --output-format csv -d "$OUT/${v}_$i" -- python3 "$ROOT/tools/prof_shade.py" $v > "$OUT/${v}_$i.log" 2>&1 || echo "set $i ($set) failed for $v"
  done
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, collections, os
for v in ("dbg0", "dbg16"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for p in glob.glob(f"gpurun_out/lds_ab/{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_material" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][(p, r["Dispatch_Id"])] += float(r["Counter_Value"])
    t = []
    for p in glob.glob(f"gpurun_out/lds_ab/{v}_1/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_material" in r["Name"]: t.append((r["Name"].split("(")[0][-40:], float(r["AverageNs"]) / 1e3))
    print({"dbg0": "register window (default)", "dbg16": "LDS tile"}[v], t, {k: round(sum(x.values()) / len(x) / 1e6, 3) for k, x in sorted(acc.items())})
PY
