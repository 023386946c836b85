#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fp16.py -m gpu -q -rf --tb=short -x 2>&1 | tail -6
timeout 300 python tools/conv_bench.py --reps 3 --tiles B,B7,B20,B13,B14 > gpurun_out/conv_sweep_bx3_r2j.txt 2>&1; cat gpurun_out/conv_sweep_bx3_r2j.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_c3_j.csv > gpurun_out/bench_c3_j.json 2> gpurun_out/bench_c3_j.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c3_j.json')); print('c3 f32', d['value'], d['engine_only'], d['roofline']['achieved'], d['roofline']['conv1x1'], d['roofline']['other_ms_per_step'])"
timeout 600 python bench.py --dtype f16 --steps 10 --warmup 2 --no-cpu-baseline --dump-ops gpurun_out/ops_c3_f16_j.csv > gpurun_out/bench_c3_f16_j.json 2> gpurun_out/bench_c3_f16_j.err; python -c "
import json; d=json.load(open('gpurun_out/bench_c3_f16_j.json')); print('c3 f16', d['value'], d['engine_only'], d['roofline']['achieved'], d['roofline']['conv1x1'], d['roofline']['other_ms_per_step'])"
