#!/bin/bash
# Build libpadel_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
# PADEL_EXTRA_FLAGS=-DPADEL_BX3_PROBES adds the (wrong-result) ceiling-probe tiles 420 / 520 of conv_tap_bx3.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value ${PADEL_EXTRA_FLAGS:-}"
mkdir -p build
pids=()
for f in conv_lds.hip conv_tap.hip conv_tap16.hip conv_tap_bx3.hip conv_patch_bx3.hip conv_tap_h2.hip conv_patch_h2.hip conv_patch16.hip kernels_misc.hip postproc.hip tracknet_post.hip; do
  [ -f "$f" ] || continue
  hipcc $FLAGS -c "$f" -o "build/${f%.hip}.o" &
  pids+=($!)
done
hipcc $FLAGS -x hip -c engine.cpp -o build/engine.o &
pids+=($!)
g++ -O2 -std=c++17 -fPIC -Wall -c bytetrack.cpp -o build/bytetrack.o &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpadel_hip.so build/*.o -Wl,-rpath,/opt/rocm/lib
echo "built $(cd .. && pwd)/libpadel_hip.so"
