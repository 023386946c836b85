set -u
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "two_product or conv_variants" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.txt
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T323,T324,T303 --reps 7 --shapes "P3.bneck,P4.bneck,P5.bneck,head0,players.P" > $O/quad_r.txt 2>&1; grep -v amdgpu.ids $O/quad_r.txt | head -20
