#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_ball.py tests/test_gpu_conv.py tests/test_gpu_runner.py -m gpu -q -rf --tb=line 2>&1 | tail -12
bash tools/pmc_bench_traffic.sh c3 2>&1 | tail -25
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_c3.log 2>&1
head -14 $R/gpurun_out/prof_c3/c3_kernel_stats.csv
