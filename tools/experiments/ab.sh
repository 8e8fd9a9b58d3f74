#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (boxes differ by a few percent): build_tmp/libA.so against the in-tree .so,
# alternating, light counts as arguments.   usage: bash tools/experiments/ab.sh 64 16 0
for rep in 1 2; do
  for lib in build_tmp/libA.so arctic-renderer_amd/csrc/libarctic_hip.so; do
    echo "== $lib"
    ARCTIC_HIP_LIBRARY=$PWD/$lib python tools/experiments/sun_only.py "$@" 2>&1 | tail -n $#
  done
done
