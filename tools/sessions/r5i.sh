set -u
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_yolo_parity.py -m gpu -q -s -x -k "fused_stem or detect_parity or stale_arena or batch_invariance" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|rel diff|Error" $O/pytest.txt | tail -20
bash tools/gpu_session.sh r5i bench_short
