set -u
O=gpurun_out/r7a; mkdir -p $O
for lib in tools/ab/libpadel_hip_earlyw.so padel_analytics_amd/libpadel_hip.so tools/ab/libpadel_hip_earlyw.so padel_analytics_amd/libpadel_hip.so; do
  echo "== $lib" | tee -a $O/h2r_latew.txt
  PADEL_LIB=$lib timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T324,T325 --reps 7 --shapes "m.P3.bneck,m.P4.bneck,m.P5.bneck,pose.P3.bneck,m.head0.P3,pose.head 192->64" 2>&1 | grep -v amdgpu.ids | head -8 | tee -a $O/h2r_latew.txt
done
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
