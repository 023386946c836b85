#!/bin/bash
# round 3, GPU call U: short GIL switch interval in the two-stage tracker loop: c4 and c3 through the runner
mkdir -p gpurun_out/r3u
for w in c4 c3; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --no-reference-default > gpurun_out/r3u/bench_$w.json 2> gpurun_out/r3u/bench_$w.err
echo "bench $w rc=$?" | tee -a gpurun_out/r3u/status.txt
python -c "
import json
d=json.load(open('gpurun_out/r3u/bench_$w.json')); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], d['engine_only']['value'], d.get('host_frames'), r['achieved'], r.get('all_kernels_ms_per_step'))
"
done
