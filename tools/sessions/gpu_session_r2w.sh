#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
T0=$(date +%s); timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/r2w_bench_c3.json 2> gpurun_out/r2w_bench_c3.err; echo "bench wall $(( $(date +%s) - T0 )) s"; grep -E "Error|Traceback" gpurun_out/r2w_bench_c3.err | head -3; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2w_bench_c3.json')); print(j['value'], j['ms_per_step']); p=j['parity']; print({k:v for k,v in p.items() if k not in ('per_tracker','bar','low_noise_heads')}); print(p.get('low_noise_heads')); print(j['cpu_baseline']); print(j['roofline']['traffic'])
PY
