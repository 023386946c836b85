#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x 2>&1 | tail -5
timeout 300 python tools/conv_bench.py --reps 3 --tiles B,B7,B20,B25,B14 --shapes "48->48,head0,48->96,16->16,32->32,P3.bneck" > gpurun_out/conv_sweep_bx3_r2h.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2h.txt
timeout 900 python -m pytest tests/test_gpu_yolo_parity.py tests/test_gpu_bench_config.py tests/test_gpu_ball.py -m gpu -q -rf --tb=line 2>&1 | tail -6
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_c3_h.csv > gpurun_out/bench_c3_h.json 2> gpurun_out/bench_c3_h.err; cat gpurun_out/bench_c3_h.json
