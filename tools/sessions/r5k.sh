set -u
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_conv.py -m gpu -q -x > $O/pytest_h2.txt 2>&1; echo "h2/conv rc=$?"; tail -2 $O/pytest_h2.txt
timeout 600 python -m pytest tests/test_gpu_yolo_parity.py -m gpu -q -s -k "outlier" > $O/pytest_outlier.txt 2>&1; echo "outlier rc=$?"; grep -E "passed|failed|worst px|Error" $O/pytest_outlier.txt | tail -6
cp gpurun_out/parity_report.json $O/ 2>/dev/null
