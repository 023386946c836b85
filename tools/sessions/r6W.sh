set -u
O=gpurun_out/r6W; mkdir -p $O
for lib in padel_analytics_amd/libpadel_hip.so tools/ab/libpadel_hip_awin4.so tools/ab/libpadel_hip_awin7.so; do
  echo "== $lib" | tee -a $O/h2s_awin.txt
  PADEL_LIB=$lib timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T245,T245:65,T245:321,T245:1089,T245:2113 --reps 7 --shapes "1x1 1152->384,pose 1x1 768->384,pose 1x1 384->384" 2>&1 | grep -v amdgpu.ids | head -5 | tee -a $O/h2s_awin.txt
done
