# decomposition of the 64x96 LDS conv tile's k-step time (PADEL_CONV_DIAG probes; results of DIAG != 0 are wrong by design)
export PADEL_CONV_LDS_VARIANT=7
for dyn in 0 120000; do
for d in 0 1 2 6 7 14 15; do
  echo "== dynlds=$dyn diag=$d"
  PADEL_CONV_DYNLDS=$dyn PADEL_CONV_DIAG=$d timeout 120 python tools/conv_bench.py --one 0 0 --reps 3 --shapes "m.P4.bneck,m.P3.bneck" 2>&1 | grep -E "RESULT" | sort -u | cut -c1-400
done
done
