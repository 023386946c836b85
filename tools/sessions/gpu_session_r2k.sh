#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -6
timeout 300 python tools/conv_bench.py --reps 3 --tiles auto,B,B7,B207,B20,B220,B6,B206,B9,B209 --shapes "bneck,head0,L3,L1" > gpurun_out/conv_sweep_bx3_r2k.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2k.txt
