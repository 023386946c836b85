#!/bin/bash
# round 3, GPU call P: quad patch kernel, requests under the first MFMA row: correctness (h2 conv variants), speed, timeline
mkdir -p gpurun_out/r3p
timeout 600 python -m pytest tests/test_gpu_h2.py -m gpu -q -k "conv_variants" > gpurun_out/r3p/pytest_h2.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3p/status.txt
tail -3 gpurun_out/r3p/pytest_h2.txt
timeout 600 python tools/conv_bench.py --dtype h2 --reps 3 --shapes "m.P4.bneck,pose.P3.bneck,m.P5.bneck,pose.head0,m.P3.bneck,m.head0" --tiles T303,T323 > gpurun_out/r3p/sweep_h2q.txt 2>&1
cat gpurun_out/r3p/sweep_h2q.txt
cp tools/probe_build/libpadel_hip.so padel_analytics_amd/libpadel_hip.so
timeout 300 python tools/timeline_probe.py --kernel h2q --out gpurun_out/r3p/timeline_h2q_192.txt > /dev/null 2> gpurun_out/r3p/timeline.err
echo "timeline rc=$?" | tee -a gpurun_out/r3p/status.txt
head -27 gpurun_out/r3p/timeline_h2q_192.txt | cut -c1-200
