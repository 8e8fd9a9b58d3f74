#!/bin/bash
# tools/experiments/shard_cost.py (whole frames and one rank of R, static sun / map redrawn) for several builds of the library, alternating, two repetitions:
#     gpurun -- 'bash tools/experiments/libs_ab.sh arctic-renderer_amd/csrc/libarctic_hip.so build_tmp/lib_x.so ... > gpurun_out/libs_ab.txt'
# RANKS="8" (default) chooses the R of shard_cost.py.
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (repetition $rep)"
    ARCTIC_HIP_LIBRARY=$PWD/$lib python3 tools/experiments/shard_cost.py ${RANKS:-8} 2>&1 | grep -v "^$" | cut -c1-200
  done
done
