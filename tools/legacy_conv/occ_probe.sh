export PADEL_CONV_LDS_VARIANT=7
for dyn in 0 24000 40000 70000 120000; do
  PADEL_CONV_DYNLDS=$dyn PADEL_CONV_OCC=1 timeout 120 python tools/conv_bench.py --one 0 0 --reps 3 --shapes "m.P4.bneck,m.P3.bneck" 2>&1 | grep -E "occ\]|RESULT" | sort -u | cut -c1-400
done
export PADEL_CONV_LDS_VARIANT=9
for dyn in 0 70000 120000; do
  PADEL_CONV_DYNLDS=$dyn PADEL_CONV_OCC=1 timeout 120 python tools/conv_bench.py --one 0 0 --reps 3 --shapes "m.P4.bneck" 2>&1 | grep -E "occ\]|RESULT" | sort -u | cut -c1-300
done
