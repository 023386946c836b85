#!/bin/bash
# head first convs merged (0) against one conv per branch (1) on the c3 graphs: engine-only step + the 3x3 rows of the heads
OUT=${1:-gpurun_out/head_ab}; mkdir -p "$OUT"
for v in 0 1; do
  PADEL_HEAD_SPLIT=$v timeout 300 python bench.py --steps 5 --warmup 2 --quick --engine-only --traffic none --dump-ops "$OUT/ops_$v.csv" > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
  echo "head_split=$v rc=$? $(python tools/bench_summary.py $OUT/bench_$v.json | head -2 | tr '\n' ' ')"
  awk -F, -v v=$v '$2==2 && $3==3 && ($6==192 || $6==384 || $6==576) && ($5==304 || $5==256 || $5==192 || $5==64 || $5==48 || $5==128) && $7==1 {printf "   split=%s %s M=%s %s->%s %sx%s %.3f ms\n", v, $1, $4, $6, $5, $8, $9, $10}' "$OUT/ops_$v.csv"
done
