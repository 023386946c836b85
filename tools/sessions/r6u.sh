set -u
O=gpurun_out/r6u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -k "two_product or conv_variants or persistent or promise" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
timeout 600 python tools/conv_bench.py --dtype h2 --w16 --tiles auto,T304,T323,T324,T325 --reps 7 --shapes "pose.head,m.head0,players.head,P4.bneck,P3.bneck" > $O/nf2_ws.txt 2>&1; grep -v amdgpu.ids $O/nf2_ws.txt | head -14
timeout 600 python tools/conv_bench.py --dtype h2 --tiles auto,T304,T325 --reps 7 --shapes "tn ,pose.head" > $O/nf2_3p.txt 2>&1; grep -v amdgpu.ids $O/nf2_3p.txt | head -10
