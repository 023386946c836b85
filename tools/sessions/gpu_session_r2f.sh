#!/bin/bash
# full validation + the numbers judged for the round (default = bf16x3 kernels)
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
( timeout 1700 python -m pytest tests -m gpu -q -rf --tb=line 2>&1 | tail -15 ) 2>&1
cp gpurun_out/parity_report.json gpurun_out/parity_report_r2.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 --dump-ops gpurun_out/ops_c3.csv > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json; tail -2 gpurun_out/bench_c3.err
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_c2.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/bench_c2.json
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --host-frames --no-roofline > gpurun_out/bench_c3_host.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3_host.json
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --graph 1 > gpurun_out/bench_c3_graph.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3_graph.json
bash tools/pmc_bench_traffic.sh c3 bx3 2>&1 | tail -16
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --engine-only --no-compare > $R/gpurun_out/prof_c3.log 2>&1
head -12 $R/gpurun_out/prof_c3/c3_kernel_stats.csv
