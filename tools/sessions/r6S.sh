set -u
O=gpurun_out/r6S; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T213,T243,T244,T249 --reps 7 --shapes "1x1 96->96,1x1 192->96,1x1 192->192,1x1 576->192" > $O/h2s_m64n96.txt 2>&1; grep -v amdgpu.ids $O/h2s_m64n96.txt | head -16
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
