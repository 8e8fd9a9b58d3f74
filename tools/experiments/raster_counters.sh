#!/bin/bash
# counters of the prepass kernels over whole frames of config 3 (each set in its own run, kernel trace only)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_raster_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_INSTS_FLAT" \
           "TCC_REQ_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/tools/frame_time.py" 3 > "$OUT/pmc$i.log" 2>&1 || echo "[raster counters] set $i failed: $set"
    echo "[raster counters] pass $i done: $set"
done
