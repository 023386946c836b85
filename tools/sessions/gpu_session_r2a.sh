#!/bin/bash
# Round-2 GPU session A: -m gpu tests, smoke, bench (runner-level + engine-only + parity), conv sweep.
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
( timeout 1700 python -m pytest tests -m gpu -q -rs --durations=8 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.txt 2>&1
tail -30 gpurun_out/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 2 --dump-ops gpurun_out/ops_c3.csv > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json; tail -5 gpurun_out/bench_c3.err
timeout 300 python tools/conv_bench.py --reps 3 --tiles auto,T6,T7,T9,T11,T13,T14,T15,T20,Ap3 > gpurun_out/conv_sweep_r2a.txt 2>&1; cat gpurun_out/conv_sweep_r2a.txt
