#!/bin/bash
# Build libpadel_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
# PADEL_EXTRA_FLAGS=-DPADEL_BX3_PROBES adds the (wrong-result) ceiling-probe tiles 420 / 520 of conv_tap_bx3.hip,
# -DPADEL_H2P_PROBES the ablation tiles 332.. of conv_patch_h2.hip, -DPADEL_STEM_PROBE=1|2|4|5 the ablations of stem_l1_h2.hip
# (profiles/r5m_stem_ablation.txt); PADEL_OUT / PADEL_BUILD_DIR keep such a build apart
# from the product library (tools only)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value ${PADEL_EXTRA_FLAGS:-}"
BUILD="${PADEL_BUILD_DIR:-build}"
OUT="${PADEL_OUT:-../libpadel_hip.so}"
mkdir -p "$BUILD"
pids=()
# PADEL_ONLY="a.hip b.hip": recompile only these translation units and relink with the objects already in $BUILD (iteration)
ALL="conv_tap.hip conv_tap16.hip conv_tap_bx3.hip conv_patch_bx3.hip conv_tap_h2.hip conv_tap_h2p.hip conv_1x1_h2s.hip conv_patch_h2.hip conv_patch_h2q.hip conv_patch_h2r.hip conv_patch_h2v.hip conv_patch_h2w.hip conv_patch16.hip kernels_misc.hip stem_l1_h2.hip postproc.hip tracknet_post.hip"
for f in ${PADEL_ONLY:-$ALL}; do
  [ -f "$f" ] || continue
  [ "$f" = engine.cpp ] && continue
  hipcc $FLAGS -c "$f" -o "$BUILD/${f%.hip}.o" &
  pids+=($!)
done
if [ -z "${PADEL_ONLY:-}" ] || [[ " $PADEL_ONLY " == *" engine.cpp "* ]]; then
hipcc $FLAGS -x hip -c engine.cpp -o "$BUILD/engine.o" &
pids+=($!)
fi
# host code; -ffp-contract=off: the tracker's doubles are pinned against the Python twin (no fused multiply-adds)
if [ -z "${PADEL_ONLY:-}" ]; then
g++ -O3 -ffp-contract=off -std=c++17 -fPIC -Wall -c bytetrack.cpp -o "$BUILD/bytetrack.o" &
pids+=($!)
fi
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$BUILD"/*.o -Wl,-rpath,/opt/rocm/lib
echo "built $OUT"
