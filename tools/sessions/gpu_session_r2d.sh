#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -rf --tb=short -x -s 2>&1 | grep -E "RMS error|passed|failed|Error|assert" | tail -30
timeout 300 python tools/conv_bench.py --reps 3 --tiles auto,B,B7,B6,B9,B20,B11,B13,B14 > gpurun_out/conv_sweep_bx3_r2d.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2d.txt
echo "=== parity suite with the bf16x3 kernels as default (PADEL_CONV_IMPL=bx3)"
PADEL_CONV_IMPL=bx3 timeout 900 python -m pytest tests/test_gpu_yolo_parity.py tests/test_gpu_bench_config.py tests/test_gpu_ball.py tests/test_gpu_runner.py -m gpu -q -rf --tb=line 2>&1 | tail -15
cp gpurun_out/parity_report.json gpurun_out/parity_report_bx3.json 2>/dev/null
echo "=== same suites, fp32 MFMA default (new stem)"
timeout 900 python -m pytest tests/test_gpu_yolo_parity.py tests/test_gpu_bench_config.py -m gpu -q -rf --tb=line 2>&1 | tail -8
PADEL_CONV_IMPL=bx3 timeout 600 python bench.py --steps 10 --warmup 2 --dump-ops gpurun_out/ops_c3_bx3.csv > gpurun_out/bench_c3_bx3.json 2> gpurun_out/bench_c3_bx3.err; cat gpurun_out/bench_c3_bx3.json
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c3_f32.json 2> gpurun_out/bench_c3_f32.err; cat gpurun_out/bench_c3_f32.json
