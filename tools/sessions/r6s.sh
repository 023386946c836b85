set -u
O=gpurun_out/r6s; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --durations=60 > $O/pytest_gpu.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.txt | tail -12
grep -A 65 "slowest" $O/pytest_gpu.txt | head -70
