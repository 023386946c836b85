set -u
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x > gpurun_out/r5c/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5c/pytest.txt
timeout 400 python tools/conv_bench.py --dtype h2 --tiles auto,T243,T213 --reps 7 --shapes s2,1x1 > gpurun_out/r5c/tiles.txt 2>&1; grep -v amdgpu.ids gpurun_out/r5c/tiles.txt | head -12
