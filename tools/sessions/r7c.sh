set -u
O=gpurun_out/r7c; mkdir -p $O
for lib in tools/ab/libpadel_hip_earlyv.so padel_analytics_amd/libpadel_hip.so tools/ab/libpadel_hip_earlyv.so padel_analytics_amd/libpadel_hip.so; do
  echo "== $lib" | tee -a $O/h2v_latew.txt
  PADEL_LIB=$lib timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto --reps 7 --shapes "P2.bneck,n.P3.bneck,n.P2.bneck" 2>&1 | grep -v amdgpu.ids | head -6 | tee -a $O/h2v_latew.txt
done
timeout 600 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
