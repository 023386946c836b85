set -u
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py -m gpu -q -x -s -k "two_product" > $O/pytest_2p.txt 2>&1; echo "two-product rc=$?"; grep -E "passed|failed|Error|RMS vs" $O/pytest_2p.txt | cut -c1-260 | tail -12
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x -k "not two_product" > $O/pytest_h2.txt 2>&1; echo "h2/conv rc=$?"; tail -2 $O/pytest_h2.txt
bash tools/gpu_session.sh r5p bench_short
