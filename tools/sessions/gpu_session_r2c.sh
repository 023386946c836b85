#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_fp16.py -m gpu -q -rf --tb=short -x 2>&1 | tail -40
timeout 300 python tools/conv_bench.py --dtype f16 --reps 3 --tiles auto,T6,T7,T9,T11,T20,T30,T31,T32 > gpurun_out/conv_sweep_f16_r2c.txt 2>&1; cat gpurun_out/conv_sweep_f16_r2c.txt
timeout 600 python bench.py --dtype f16 --steps 10 --warmup 2 --dump-ops gpurun_out/ops_c3_f16.csv > gpurun_out/bench_c3_f16.json 2> gpurun_out/bench_c3_f16.err; cat gpurun_out/bench_c3_f16.json; tail -3 gpurun_out/bench_c3_f16.err
