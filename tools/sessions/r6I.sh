set -u
O=gpurun_out/r6I; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T243,T244:9,T244,T245:9,T245 --reps 7 --shapes "pose 1x1,1x1 1152,1x1 192->192,1x1 576->192" > $O/h2s_window.txt 2>&1; grep -v amdgpu.ids $O/h2s_window.txt | head -16
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
