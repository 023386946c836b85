#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_bench_config.py -m gpu -q -rf --tb=short -x 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-compare > gpurun_out/r2zb_bench_c3.json 2> gpurun_out/r2zb_bench_c3.err; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2zb_bench_c3.json')); print(j['value'], j['ms_per_step'], j['engine_only']['value']); r=j['roofline']; print({k:r[k] for k in ('achieved','kernel_ms_per_step','all_kernels_ms_per_step','other_ms_per_step')})
PY
