#!/bin/bash
# round 3, GPU call A: first contact of the h2 kernels — unit tests, conv sweep, short bench
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_h2.py -q --maxfail=12 -s > gpurun_out/r3a/pytest_h2.txt 2>&1
echo "pytest h2 rc=$?" | tee -a gpurun_out/r3a/status.txt
tail -40 gpurun_out/r3a/pytest_h2.txt
timeout 400 python tools/conv_bench.py --dtype h2 --tiles auto,T220,T213,T209,T207,T303,T304,T306 --reps 3 > gpurun_out/r3a/conv_h2_sweep.txt 2>&1
echo "sweep h2 rc=$?" | tee -a gpurun_out/r3a/status.txt
timeout 300 python tools/conv_bench.py --dtype f32 --tiles B,B303,B213 --reps 3 > gpurun_out/r3a/conv_bx3_sweep.txt 2>&1
echo "sweep bx3 rc=$?" | tee -a gpurun_out/r3a/status.txt
cat gpurun_out/r3a/conv_h2_sweep.txt gpurun_out/r3a/conv_bx3_sweep.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-ops gpurun_out/r3a/ops_c3_h2.csv > gpurun_out/r3a/bench_c3_h2.json 2> gpurun_out/r3a/bench_c3_h2.err
echo "bench rc=$?" | tee -a gpurun_out/r3a/status.txt
cat gpurun_out/r3a/bench_c3_h2.json; tail -5 gpurun_out/r3a/bench_c3_h2.err
