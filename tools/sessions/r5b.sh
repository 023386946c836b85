set -u
mkdir -p gpurun_out/r5b
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "conv_variants or upsample_absorbed" > gpurun_out/r5b/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5b/pytest.txt
timeout 400 python tools/conv_bench.py --dtype h2 --tiles auto,T243,T213,T239,T209 --reps 7 --shapes 1x1 > gpurun_out/r5b/tiles_1x1.txt 2>&1; grep -v amdgpu.ids gpurun_out/r5b/tiles_1x1.txt | head -12
