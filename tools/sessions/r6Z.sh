set -u
O=gpurun_out/r6Z2; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T245,T247,T251 --reps 7 --shapes "pose 1x1,1x1 1152,1x1 192->192,1x1 576->192" > $O/h2s_m32.txt 2>&1; grep -v amdgpu.ids $O/h2s_m32.txt | head -16
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
