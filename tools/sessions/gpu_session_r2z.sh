#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -rf --tb=short 2>&1 | tail -6 | tee gpurun_out/fin2_pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/fin2_smoke.txt
timeout 400 python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/fin2_bench_c4.json 2> gpurun_out/fin2_bench_c4.err; cut -c1-200 gpurun_out/fin2_bench_c4.json
timeout 900 python bench.py > gpurun_out/fin2_bench_default.json 2> gpurun_out/fin2_bench_default.err; cut -c1-200 gpurun_out/fin2_bench_default.json
