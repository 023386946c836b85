#!/bin/bash
# round 3, GPU call H: (1) the tests added after call G (arena poison, fp16 saturation), (2) ablation ceilings of the h2 patch
# kernel (library built with -DPADEL_H2P_PROBES; WRONG results by construction, timing only)
mkdir -p gpurun_out/r3h
timeout 600 python -m pytest tests/test_gpu_h2.py tests/test_gpu_fp16.py -m gpu -q > gpurun_out/r3h/pytest_new.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3h/status.txt
tail -3 gpurun_out/r3h/pytest_new.txt
cp padel_analytics_amd/libpadel_hip.so /tmp/product.so
cp tools/probe_build/libpadel_hip.so padel_analytics_amd/libpadel_hip.so
T=T303,T333,T334,T335,T336,T337,T339,T340,T342,T347,T348,T349,T363
timeout 600 python tools/conv_bench.py --dtype h2 --reps 3 --shapes "m.P4.bneck,pose.P3.bneck,m.P5.bneck,pose.head0" --tiles $T > gpurun_out/r3h/ablate_h2p.txt 2>&1
echo "ablate rc=$?" | tee -a gpurun_out/r3h/status.txt
cat gpurun_out/r3h/ablate_h2p.txt
cp /tmp/product.so padel_analytics_amd/libpadel_hip.so
