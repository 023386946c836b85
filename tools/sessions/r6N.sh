set -u
O=gpurun_out/r6N; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T213,T246,T248 --reps 7 --shapes "s2 ,m.L3" > $O/h2s3_m64.txt 2>&1; grep -v amdgpu.ids $O/h2s3_m64.txt | head -16
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; tail -3 $O/pytest_h2.txt
