#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -15
T="B,B213,B220,B207"; S="1x1,L3,L1,m.P2"
timeout 300 python tools/conv_bench.py --reps 3 --tiles $T --shapes "$S" > gpurun_out/conv_sweep_bx3_r2s_new.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2s_new.txt
cp padel_analytics_amd/libpadel_hip.so /tmp/new.so; cp padel_analytics_amd/libpadel_hip_prev.so padel_analytics_amd/libpadel_hip.so
timeout 300 python tools/conv_bench.py --reps 3 --tiles $T --shapes "$S" > gpurun_out/conv_sweep_bx3_r2s_prev.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2s_prev.txt
cp /tmp/new.so padel_analytics_amd/libpadel_hip.so
timeout 300 python tools/conv_bench.py --reps 3 --tiles $T --shapes "$S" > gpurun_out/conv_sweep_bx3_r2s_new2.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2s_new2.txt
