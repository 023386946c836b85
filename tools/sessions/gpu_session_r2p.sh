#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -15
timeout 300 python tools/conv_bench.py --reps 3 --tiles B,B220,B303,B313,B304,B314,B213,B209 --shapes "m.P,pose.P,head0,1x1" > gpurun_out/conv_sweep_bx3_r2p.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2p.txt
