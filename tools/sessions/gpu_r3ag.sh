#!/bin/bash
# round 3, GPU call AG: SPPF pools with clamped windows: helper-kernel tests, graph-level tests, per-op times
mkdir -p gpurun_out/r3ag
timeout 600 python -m pytest tests/test_gpu_h2.py tests/test_gpu_bench_config.py tests/test_gpu_fp16.py tests/test_gpu_conv.py -m gpu -q -k "not conv_variants and not conv16" > gpurun_out/r3ag/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3ag/pytest.txt
timeout 600 python bench.py --engine-only --no-compare --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3ag/ops_c3.csv > gpurun_out/r3ag/bench.json 2> gpurun_out/r3ag/bench.err
grep -E "^(players|ball|pose),3," gpurun_out/r3ag/ops_c3.csv
python -c "
import json; d=json.load(open('gpurun_out/r3ag/bench.json')); print(d['engine_only']['value'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['other_ms_per_step'])"
