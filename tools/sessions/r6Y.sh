set -u
O=gpurun_out/r6Y; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T245,T247,T250,T250:65 --reps 7 --shapes "pose 1x1,1x1 1152" > $O/h2s_n384.txt 2>&1; grep -v amdgpu.ids $O/h2s_n384.txt | head -16
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
