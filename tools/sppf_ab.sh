#!/bin/bash
# SPPF pooling time (profile rows of kind 3) of the three c3 graphs: three pool launches (0) against the fused kernel at
# 1024 / 256 / 512 threads (1 / 2 / 3)
OUT=${1:-gpurun_out/sppf_ab}; mkdir -p "$OUT"
for v in 0 1 2 3; do
  PADEL_FUSE_SPPF=$v timeout 300 python bench.py --steps 2 --warmup 1 --quick --engine-only --traffic none --dump-ops "$OUT/ops_$v.csv" > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
  echo "fuse_sppf=$v rc=$? $(grep ',3,5,' "$OUT/ops_$v.csv" | awk -F, '{printf "%s %.3f ms  ", $1, $10}')"
done
