#!/bin/bash
# whole-frame A/B of two builds in one gpurun call: build_tmp/libA.so against the in-tree library
for rep in 1 2; do
  for lib in build_tmp/libA.so arctic-renderer_amd/csrc/libarctic_hip.so; do
    echo "== $lib"
    ARCTIC_HIP_LIBRARY=$PWD/$lib python tools/frame_time.py "$@" 2>&1 | grep "whole frame"
  done
done
