set -u
O=gpurun_out/r6v; mkdir -p $O
timeout 600 python tools/tracknet_bench.py --frames 264 --feed 64 --dump-ops $O/tracknet_ops.csv > $O/tracknet_bench.json 2> $O/tracknet_bench.err; echo "tracknet rc=$?"; cat $O/tracknet_bench.json
bash tools/gpu_session.sh r6v bench_short
timeout 900 python -m pytest tests/test_gpu_ball.py tests/test_gpu_h2.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.txt
