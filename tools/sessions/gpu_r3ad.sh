#!/bin/bash
# round 3, GPU call AD: the suites not re-run since the wide kernel entered the automatic choice
mkdir -p gpurun_out/r3ad
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fp16.py tests/test_gpu_baseline_configs.py tests/test_gpu_runner.py -m gpu -q > gpurun_out/r3ad/pytest_rest.txt 2>&1
echo "pytest rest rc=$?" | tee -a gpurun_out/r3ad/status.txt
tail -3 gpurun_out/r3ad/pytest_rest.txt
