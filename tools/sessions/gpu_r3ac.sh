#!/bin/bash
# round 3, GPU call AC: the final kernel set — full default bench line, c2, driver command line, smoke, the conv / fp16 / baseline-config suites
mkdir -p gpurun_out/r3ac
timeout 1200 python bench.py --dump-ops gpurun_out/r3ac/ops_c3.csv > gpurun_out/r3ac/bench_c3.json 2> gpurun_out/r3ac/bench_c3.err
echo "bench c3 rc=$?" | tee -a gpurun_out/r3ac/status.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-frames --no-reference-default > gpurun_out/r3ac/bench_c3_driver_cmdline.json 2> gpurun_out/r3ac/bench_c3_driver.err
timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-host-frames > gpurun_out/r3ac/bench_c2.json 2> gpurun_out/r3ac/bench_c2.err
python -c "
import json
for n in ('c3','c3_driver_cmdline','c2'):
    d=json.load(open('gpurun_out/r3ac/bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['engine_only']['value'], d['roofline']['achieved'], d['roofline']['frac'], d.get('host_frames',{}).get('fanout_frames_per_s'), d.get('reference_default',{}).get('value'))
"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3ac/smoke.txt 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/r3ac/status.txt
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fp16.py tests/test_gpu_baseline_configs.py tests/test_gpu_preprocess.py -m gpu -q > gpurun_out/r3ac/pytest_rest.txt 2>&1
echo "pytest rest rc=$?" | tee -a gpurun_out/r3ac/status.txt
tail -3 gpurun_out/r3ac/pytest_rest.txt
