#!/bin/bash
# round 3, GPU call AB: graph-level tests after the tap-order change + quad kernel in the auto choice; c3 bench line
mkdir -p gpurun_out/r3ab
timeout 900 python -m pytest tests/test_gpu_h2.py tests/test_gpu_yolo_parity.py tests/test_gpu_bench_config.py tests/test_gpu_known_answers.py tests/test_gpu_ball.py -m gpu -q > gpurun_out/r3ab/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3ab/status.txt
tail -5 gpurun_out/r3ab/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-host-frames --no-reference-default --dump-ops gpurun_out/r3ab/ops_c3.csv > gpurun_out/r3ab/bench_c3.json 2> gpurun_out/r3ab/bench_c3.err
echo "bench rc=$?" | tee -a gpurun_out/r3ab/status.txt
python -c "
import json
d=json.load(open('gpurun_out/r3ab/bench_c3.json')); print(d['value'], d['ms_per_step'], d['engine_only']['value'], d['roofline'])
"
