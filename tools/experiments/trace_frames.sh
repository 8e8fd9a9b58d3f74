#!/bin/bash
# kernel trace of whole frames: per-kernel averages and a timeline of a window of dispatches.  On the GPU box, from the repo root:
#   bash tools/experiments/trace_frames.sh <out dir under gpurun_out> <first dispatch> <count> [SHARD=band_rows,index,count] [option=value ...]
R=$(pwd); OUT=$R/gpurun_out/$1; FIRST=$2; COUNT=$3; shift 3
mkdir -p $OUT
if [[ "$1" == SHARD=* ]]; then export "$1"; shift; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -- python3 $R/tools/experiments/frames_for_trace.py 3 60 "$@" > $OUT/run.log 2>&1
cd $R
python tools/kernel_times.py $OUT/tr > $OUT/kernels.txt
python tools/experiments/timeline.py $(find $OUT/tr -name "*kernel_trace.csv") $FIRST $COUNT > $OUT/timeline.txt
rm -rf $OUT/tr
cat $OUT/kernels.txt
