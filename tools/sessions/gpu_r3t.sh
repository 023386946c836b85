#!/bin/bash
# round 3, GPU call T: fp16 quad kernel with double-buffered tap weights; c4 bench line
mkdir -p gpurun_out/r3t
timeout 600 python -m pytest tests/test_gpu_fp16.py -m gpu -q -x -k conv16 > gpurun_out/r3t/pytest_fp16.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3t/status.txt
tail -3 gpurun_out/r3t/pytest_fp16.txt
timeout 600 python tools/conv_bench.py --dtype f16 --reps 3 --shapes "m.P4.bneck,pose.P3.bneck,pose.head0,m.P3.bneck,m.head0" --tiles auto,T304,T306,T326 > gpurun_out/r3t/sweep_p16q.txt 2>&1
cat gpurun_out/r3t/sweep_p16q.txt
timeout 600 python bench.py --workload c4 --no-cpu-baseline --no-host-frames --dump-ops gpurun_out/r3t/ops_c4.csv > gpurun_out/r3t/bench_c4.json 2> gpurun_out/r3t/bench_c4.err
echo "bench c4 rc=$?" | tee -a gpurun_out/r3t/status.txt
python -c "
import json
d=json.load(open('gpurun_out/r3t/bench_c4.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['engine_only']['value'], r['achieved'], r['frac'], r.get('kernel_ms_per_step'), r.get('conv1x1'), r.get('all_kernels_ms_per_step'), r.get('other_ms_per_step'))
"
