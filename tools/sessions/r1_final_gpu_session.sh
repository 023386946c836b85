#!/bin/bash
# One GPU session that produces everything judged for a round: -m gpu tests, smoke, bench lines, rocprofv3 summary.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 5 --warmup 2 --dump-ops gpurun_out/ops_c3.csv > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json
timeout 300 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c2.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/bench_c2.json
timeout 300 python bench.py --scales pose=n --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c3_pose_n.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3_pose_n.json
timeout 300 python tools/tracknet_bench.py > gpurun_out/tracknet_bench.json 2>> gpurun_out/bench_c3.err; cat gpurun_out/tracknet_bench.json
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_c3.log 2>&1
head -12 $R/gpurun_out/prof_c3/c3_kernel_stats.csv
