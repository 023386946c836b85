#!/bin/bash
# Round-2 closing GPU session: -m gpu tests, smoke, bench lines (c3 default with cpu_baseline + parity, c2, fp16, c4,
# host frames), rocprofv3 kernel stats of the bench command, PMC traffic + MFMA utilisation, TrackNet tracker rate.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -rf --tb=short 2>&1 | tail -6 | tee gpurun_out/fin_pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/fin_smoke.txt
timeout 900 python bench.py --steps 10 --warmup 2 --dump-ops gpurun_out/fin_ops_c3.csv > gpurun_out/fin_bench_c3.json 2> gpurun_out/fin_bench_c3.err; cut -c1-300 gpurun_out/fin_bench_c3.json
timeout 300 python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/fin_bench_c2.json 2> gpurun_out/fin_bench_c2.err; cut -c1-200 gpurun_out/fin_bench_c2.json
timeout 400 python bench.py --dtype f16 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/fin_bench_c3_f16.json 2> gpurun_out/fin_bench_c3_f16.err; cut -c1-200 gpurun_out/fin_bench_c3_f16.json
timeout 400 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/fin_bench_c4.json 2> gpurun_out/fin_bench_c4.err; cut -c1-200 gpurun_out/fin_bench_c4.json
timeout 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-compare --host-frames > gpurun_out/fin_bench_c3_host.json 2> gpurun_out/fin_bench_c3_host.err; cut -c1-200 gpurun_out/fin_bench_c3_host.json
timeout 300 python tools/tracknet_bench.py > gpurun_out/fin_tracknet_bench.json 2> gpurun_out/fin_tracknet.err; cat gpurun_out/fin_tracknet_bench.json
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin_prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/fin_prof_c3.log 2>&1 )
head -8 $(find gpurun_out/fin_prof_c3 -name "*kernel_stats.csv" | head -1) | cut -c1-160
bash tools/pmc_bench_traffic.sh c3 bx3 2>&1 | tail -14
bash tools/pmc_bx3.sh 2>&1 | tail -12 | tee gpurun_out/fin_pmc_bx3.txt
