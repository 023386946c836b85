set -u
O=gpurun_out/r5h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "conv_variants" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
timeout 400 python tools/conv_bench.py --dtype h2 --tiles auto,T243,T241,T213 --reps 7 --shapes 1x1 > $O/tiles.txt 2>&1; grep -v amdgpu.ids $O/tiles.txt | head -8
