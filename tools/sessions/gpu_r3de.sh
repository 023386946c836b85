#!/bin/bash
# round 3, GPU call D: new tests (BASELINE configs[3] / [4], known answers, h2 units after the epilogue change), conv sweep,
# the full default bench line (host_frames, reference_default, objects_materialised)
mkdir -p gpurun_out/r3e
timeout 1200 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_h2.py tests/test_gpu_known_answers.py -q -s --maxfail=10 > gpurun_out/r3e/pytest_new.txt 2>&1
echo "pytest new rc=$?" | tee -a gpurun_out/r3e/status.txt
grep -E "passed|failed|FAILED|configs\[4\]" gpurun_out/r3e/pytest_new.txt | tail -12
timeout 300 python tools/conv_bench.py --dtype h2 --tiles auto,T303,T313,T304,T314 --reps 4 > gpurun_out/r3e/conv_h2_sweep.txt 2>&1
echo "sweep rc=$?" | tee -a gpurun_out/r3e/status.txt
cat gpurun_out/r3e/conv_h2_sweep.txt
timeout 1200 python bench.py --dump-ops gpurun_out/r3e/ops_c3.csv > gpurun_out/r3e/bench_c3.json 2> gpurun_out/r3e/bench_c3.err
echo "bench rc=$?" | tee -a gpurun_out/r3e/status.txt
cat gpurun_out/r3e/bench_c3.json; tail -5 gpurun_out/r3e/bench_c3.err
