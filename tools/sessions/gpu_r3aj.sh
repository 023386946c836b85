#!/bin/bash
# round 3, GPU call AJ: c4 and c2 lines with the final build
mkdir -p gpurun_out/r3aj
timeout 300 python bench.py --workload c4 --no-cpu-baseline --no-host-frames > gpurun_out/r3aj/bench_c4.json 2> gpurun_out/r3aj/bench_c4.err
timeout 300 python bench.py --workload c2 --no-cpu-baseline --no-host-frames > gpurun_out/r3aj/bench_c2.json 2> gpurun_out/r3aj/bench_c2.err
python -c "
import json
for n in ('c4','c2'):
    d=json.load(open('gpurun_out/r3aj/bench_%s.json'%n)); r=d['roofline']; print(n, d['value'], d['ms_per_step'], d['engine_only']['value'], r['achieved'], r['frac'], r['conv1x1']['achieved'], r['all_kernels_ms_per_step'], r['other_ms_per_step'])
"
