#!/bin/bash
# round 3, GPU call AO (3-stage weight ring): fused stem + layer 1: the test asserts the fused kernel really ran; its effect on the c3 step (engine-only)
mkdir -p gpurun_out/r3ao
timeout 100 python -m pytest tests/test_gpu_h2.py -m gpu -q -k fused_stem > gpurun_out/r3ao/pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3ao/pytest.txt | cut -c1-200
PADEL_FUSE_STEM=1 timeout 200 python bench.py --engine-only --no-compare --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3ao/ops_c3_fused.csv > gpurun_out/r3ao/bench_fused.json 2> gpurun_out/r3ao/bench_fused.err
grep -E "^(players|ball|pose),1," gpurun_out/r3ao/ops_c3_fused.csv
python -c "
import json; d=json.load(open('gpurun_out/r3ao/bench_fused.json')); print(d['engine_only']['value'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['other_ms_per_step'])"
