set -u
O=gpurun_out/r6H; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_yolo_parity.py -m gpu -q -x -s -k least_squares ) > $O/pytest_fit.txt 2>&1; grep -E "least-squares heads|passed|failed|Error|real" $O/pytest_fit.txt | cut -c1-900 | tail -12
cp gpurun_out/parity_report.json $O/parity_report_fit.json 2>/dev/null
