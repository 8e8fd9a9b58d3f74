#!/bin/bash
# The rocprofv3 recipe behind profiles/<tag>_* and profiles/pmc_latest.json; run on the GPU box from the repo root:
#     gpurun -- 'bash tools/profile_round.sh r2_a'
# Pass 1: kernel trace + stats of the bench command itself (no extras, no CPU baseline: its k_material average is the timed
# launches').  Passes 2-6: counters, each in its own run with --kernel-trace only (never with other trace domains), over
# tools/prof_shade.py (5 launches of the config-3 shading pass).  FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC has 4 slots).
set -e -o pipefail
TAG=${1:?usage: profile_round.sh <tag>}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/bench.py" --no-extras --no-cpu --steps 100 > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "[profile] kernel trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --stats --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/prof_shade.py" full > "$OUT/pmc$i.log" 2>&1
    echo "[profile] counters pass $i done: $set"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" "$TAG"
cp "$OUT"/kt/*/*kernel_stats.csv "profiles/${TAG}_kernel_stats_bench_command.csv" 2>/dev/null || cp $(find "$OUT/kt" -name "*kernel_stats.csv" | head -1) "profiles/${TAG}_kernel_stats_bench_command.csv"
cp "$OUT/bench.json" "profiles/${TAG}_bench_under_rocprof.json"
# the same trace cut to the K timed launches of that run (the stats file above averages every launch of the process: cold region, settle
# loop, isolated launches): the figure that has to agree with the line's ms_per_step
python3 tools/kernel_times.py "$OUT/kt" --timed "$OUT/bench.json" > "profiles/${TAG}_kernel_trace_timed_launches.json"
cat "profiles/${TAG}_kernel_trace_timed_launches.json"
