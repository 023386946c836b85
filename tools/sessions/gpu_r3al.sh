#!/bin/bash
# round 3, GPU call AL: runner tests (fan-out path) after the try / finally refactor
mkdir -p gpurun_out/r3al
timeout 300 python -m pytest tests/test_gpu_runner.py -m gpu -q > gpurun_out/r3al/pytest.txt 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/r3al/pytest.txt | tail -2
