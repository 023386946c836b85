set -u
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
timeout 900 python tools/conv_bench.py --dtype h2 --w16 --tiles auto --reps 7 > $O/sweep_w16.txt 2>&1; grep -v amdgpu.ids $O/sweep_w16.txt | head -32
bash tools/gpu_session.sh r6w bench_short
