#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -5
timeout 300 python tools/conv_bench.py --reps 3 --tiles B,B7,B107,B20,B120,B11,B111,B13,B113,B14,B114 > gpurun_out/conv_sweep_bx3_r2e.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2e.txt
