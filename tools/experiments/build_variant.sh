#!/bin/bash
# A variant build of the library for same-box A/B runs: shade.hip (and geometry.hip when GEO=1) compiled with extra -D switches, linked with the
# in-tree objects into build_tmp/lib_<name>.so (git-ignored; travels to the GPU box).   usage: bash tools/experiments/build_variant.sh <name> [-DX=1 ...]
set -e
NAME=${1:?name}; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/arctic-renderer_amd/csrc
mkdir -p "$ROOT/build_tmp/obj_$NAME"
O=$ROOT/build_tmp/obj_$NAME
make -s -C "$SRC"
COMMON="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950"
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -mllvm -disable-machine-licm "$@" -c "$SRC/shade.hip" -o "$O/shade.o"
GEOOBJ=$SRC/geometry.o
if [ "${GEO:-0}" = "1" ]; then /opt/rocm/bin/hipcc $COMMON -ffp-contract=off "$@" -c "$SRC/geometry.hip" -o "$O/geometry.o"; GEOOBJ=$O/geometry.o; fi
RENOBJ=$SRC/renderer.o
if [ "${REN:-0}" = "1" ]; then /opt/rocm/bin/hipcc $COMMON "$@" -c "$SRC/renderer.cpp" -o "$O/renderer.o"; RENOBJ=$O/renderer.o; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_tmp/lib_$NAME.so" "$GEOOBJ" "$O/shade.o" "$SRC/exchange.o" "$RENOBJ" "$SRC/host_math.o"
echo "built build_tmp/lib_$NAME.so"
