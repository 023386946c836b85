set -u
O=gpurun_out/r6x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T343,T343:9,T342,T342:9,T341,T341:9 --reps 7 --shapes "P2.bneck,n.P3,n.P2" > $O/h2v.txt 2>&1; grep -v amdgpu.ids $O/h2v.txt | head -10
