set -u
O=gpurun_out/r6l; mkdir -p $O
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles T323,T324,T324:9,T324:17 --reps 7 --shapes "P3.bneck,P4.bneck,P5.bneck,head0,players.P" > $O/phase.txt 2>&1; grep -v amdgpu.ids $O/phase.txt | head -14
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "two_product or conv_variants or persistent" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
