set -u
O=gpurun_out/r5g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_yolo_parity.py -m gpu -q -s -k "outlier or pose_parity or ball_n_nc1" > $O/pytest2.txt 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error|ratios|worst h2|head maps:" $O/pytest2.txt | cut -c1-600 | tail -30
cp gpurun_out/parity_report.json $O/parity_report2.json 2>/dev/null
