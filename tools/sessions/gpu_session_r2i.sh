#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_fp16.py -m gpu -q -rf --tb=short -x -k conv16 2>&1 | tail -8
timeout 300 python tools/conv_bench.py --dtype f16 --reps 3 --tiles auto,T7,T47,T31,T71,T30,T70,T9,T49,T20,T60,T46,T72 > gpurun_out/conv_sweep_f16_r2i.txt 2>&1; cat gpurun_out/conv_sweep_f16_r2i.txt
