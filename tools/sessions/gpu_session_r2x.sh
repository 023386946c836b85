#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_fp16.py -m gpu -q -rf --tb=short -x -k "conv16" 2>&1 | tail -15
timeout 300 python tools/conv_bench.py --dtype f16 --reps 3 --tiles auto,T303,T304,T306,T31,T49,T72 --shapes "m.P,pose.P,head0" > gpurun_out/conv_sweep_f16_r2x.txt 2>&1; cat gpurun_out/conv_sweep_f16_r2x.txt
