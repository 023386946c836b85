#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -5
timeout 300 python tools/conv_bench.py --reps 3 --tiles B,B7,B20,B21,B22,B23,B24,B13,B14 > gpurun_out/conv_sweep_bx3_r2g.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2g.txt
