#!/bin/bash
# round 3, GPU call B: the whole GPU suite with h2 as the default arithmetic, parity report, default bench, TrackNet bench
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -s > gpurun_out/r3b/pytest_gpu.txt 2>&1
echo "pytest gpu rc=$?" | tee -a gpurun_out/r3b/status.txt
grep -E "passed|failed|FAILED" gpurun_out/r3b/pytest_gpu.txt | tail -20
cp gpurun_out/parity_report.json gpurun_out/r3b/parity_report.json 2>/dev/null
timeout 900 python bench.py --dump-ops gpurun_out/r3b/ops_c3.csv > gpurun_out/r3b/bench_c3.json 2> gpurun_out/r3b/bench_c3.err
echo "bench rc=$?" | tee -a gpurun_out/r3b/status.txt
cat gpurun_out/r3b/bench_c3.json
timeout 300 python tools/tracknet_bench.py > gpurun_out/r3b/tracknet_bench.json 2> gpurun_out/r3b/tracknet_bench.err
echo "tracknet rc=$?" | tee -a gpurun_out/r3b/status.txt
cat gpurun_out/r3b/tracknet_bench.json
