set -u
O=gpurun_out/r6C; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "two_product or conv_variants" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T243,T244,T245,T245:65 --reps 7 --shapes "1x1" > $O/h2s_wide.txt 2>&1; grep -v amdgpu.ids $O/h2s_wide.txt | head -10
