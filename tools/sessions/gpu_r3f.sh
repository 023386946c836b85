#!/bin/bash
# round 3, GPU call F: whole GPU suite (new decode / NMS kernels, pipelined patch tiles 313 / 314), conv sweep, bench
mkdir -p gpurun_out/r3f
timeout 1800 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r3f/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r3f/status.txt
grep -E "passed|failed|FAILED" gpurun_out/r3f/pytest_gpu.txt | tail -12
cp gpurun_out/parity_report.json gpurun_out/r3f/parity_report.json 2>/dev/null
timeout 300 python tools/conv_bench.py --dtype h2 --tiles auto,T303,T313,T304,T314 --reps 4 > gpurun_out/r3f/conv_h2_sweep.txt 2>&1
echo "sweep rc=$?" | tee -a gpurun_out/r3f/status.txt
cat gpurun_out/r3f/conv_h2_sweep.txt
timeout 1200 python bench.py --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3f/ops_c3.csv > gpurun_out/r3f/bench_c3.json 2> gpurun_out/r3f/bench_c3.err
echo "bench rc=$?" | tee -a gpurun_out/r3f/status.txt
cat gpurun_out/r3f/bench_c3.json; tail -3 gpurun_out/r3f/bench_c3.err
