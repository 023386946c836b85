#!/bin/bash
# round 3, GPU call C: known-answer GPU tests, PMC profile of the h2 kernels, HBM traffic + rocprofv3 --stats of the bench
mkdir -p gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_known_answers.py -q -s > gpurun_out/r3c/pytest_known.txt 2>&1
echo "pytest known rc=$?" | tee -a gpurun_out/r3c/status.txt; tail -15 gpurun_out/r3c/pytest_known.txt
bash tools/pmc_h2.sh > gpurun_out/r3c/pmc_h2.txt 2>&1
echo "pmc_h2 rc=$?" | tee -a gpurun_out/r3c/status.txt; cat gpurun_out/r3c/pmc_h2.txt
bash tools/pmc_bench_traffic.sh c3 h2 > gpurun_out/r3c/traffic.txt 2>&1
echo "traffic rc=$?" | tee -a gpurun_out/r3c/status.txt; tail -20 gpurun_out/r3c/traffic.txt
cp profiles/r3_traffic.json gpurun_out/r3c/r3_traffic.json 2>/dev/null
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3c/stats -o s -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r3c/bench_under_rocprof.json 2> $R/gpurun_out/r3c/bench_under_rocprof.err
echo "stats rc=$?" | tee -a $R/gpurun_out/r3c/status.txt
cd $R; find gpurun_out/r3c/stats -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/r3c/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r3c/kernel_stats.csv && head -25 gpurun_out/r3c/kernel_stats.csv
# keep the merge-back small: drop the raw traces
find gpurun_out/r3c/stats -name "*kernel_trace.csv" -delete; rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE gpurun_out/pmch1 gpurun_out/pmch2 gpurun_out/pmch3
