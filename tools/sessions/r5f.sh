set -u
O=gpurun_out/r5f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k "two_tickets or pipelined_equals" > $O/pytest_a.txt 2>&1; echo "A (without the new test) rc=$?"; tail -2 $O/pytest_a.txt
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -s -k "close_with or pipelined_equals" > $O/pytest_b.txt 2>&1; echo "B (new test + next) rc=$?"; grep -v "^  File\|^Extension" $O/pytest_b.txt | tail -12
