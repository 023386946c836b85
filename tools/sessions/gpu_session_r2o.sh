#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_bench_config.py -m gpu -q -rf --tb=short -x 2>&1 | tail -6
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r2o_bench_c3.json 2> gpurun_out/r2o_bench_c3.err; cat gpurun_out/r2o_bench_c3.json
