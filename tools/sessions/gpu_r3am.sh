#!/bin/bash
# round 3, GPU call AM: the fused stem + layer-1 kernel (opt-in "fuse_stem"): bitwise against the unfused path; its time in c3
mkdir -p gpurun_out/r3am
timeout 150 python -m pytest tests/test_gpu_h2.py -m gpu -q -k fused_stem > gpurun_out/r3am/pytest.txt 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r3am/pytest.txt | cut -c1-220
