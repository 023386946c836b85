#!/bin/bash
# round 3, GPU call J: h2 unit tests after the tap-order change (tail pairing fix), timeline of the quad patch kernel
mkdir -p gpurun_out/r3j
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q > gpurun_out/r3j/pytest_h2.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3j/status.txt
tail -8 gpurun_out/r3j/pytest_h2.txt
cp tools/probe_build/libpadel_hip.so padel_analytics_amd/libpadel_hip.so
timeout 300 python tools/timeline_probe.py --kernel h2q --out gpurun_out/r3j/timeline_h2q_192.txt > /dev/null 2> gpurun_out/r3j/timeline.err
echo "timeline rc=$?" | tee -a gpurun_out/r3j/status.txt
head -60 gpurun_out/r3j/timeline_h2q_192.txt | cut -c1-200
tail -5 gpurun_out/r3j/timeline.err
